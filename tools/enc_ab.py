"""Encoder A/B: time of anx_batch_encode_packed alone and of the encode + run loop (a fresh batch per step) on the bench workload.
usage: ANX_LIB=... python3 tools/enc_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import analiticcl_amd as A
from analiticcl_amd import synth
p = synth.materialize_golden("/tmp/anxdata")
g = A.VariantModel(p["alphabet"], A.Weights(), device=0); g.read_lexicon(p["eng"]); g.build()
qs = synth.make_queries(synth.load_lexicon_words(p["eng"]), 1_000_000, max_len=16, seed=synth.SEED)
params = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
packed = ("\0".join(qs) + "\0").encode()
for _ in range(3):
    b = g.encode_packed(packed, len(qs), params); b.run(); b.free()
t = time.perf_counter()
for _ in range(10):
    b = g.encode_packed(packed, len(qs), params); b.free()
enc = (time.perf_counter() - t) / 10
live = []
def step():
    b = g.encode_packed(packed, len(qs), params); b.run_async(0); live.append(b)
    if len(live) > 1:
        o = live.pop(0); o.wait(); o.free()
for _ in range(3): step()
t = time.perf_counter()
for _ in range(20): step()
while live:
    o = live.pop(0); o.wait(); o.free()
loop = (time.perf_counter() - t) / 20
print(f"encode alone {enc * 1e3:.3f} ms, encode + run loop {loop * 1e3:.3f} ms per batch")
