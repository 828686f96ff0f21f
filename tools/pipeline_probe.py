"""Stage times of anx_pipeline on the bench workload (ANX_ENCODE_TIMING=1 prints them): pipeline_probe.py [depth] [jobs]"""
import os, sys, time
os.environ["ANX_ENCODE_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import analiticcl_amd as A
from analiticcl_amd import synth
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
p = synth.materialize_golden("/tmp/anxdata")
g = A.VariantModel(p["alphabet"], A.Weights(), device=0); g.read_lexicon(p["eng"]); g.build()
qs = synth.make_queries(synth.load_lexicon_words(p["eng"]), 1_000_000, max_len=16, seed=synth.SEED)
params = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
packed = ("\0".join(qs) + "\0").encode()
pl = A.Pipeline(g, depth=depth)
for _ in range(depth): pl.submit(packed, len(qs), params)
for _ in range(depth): pl.next()
sys.stderr.write("---- timed ----\n")
t = time.perf_counter(); sub = 0
for k in range(jobs):
    pl.submit(packed, len(qs), params); sub += 1
    if sub >= depth: pl.next(); sub -= 1
while sub: pl.next(); sub -= 1
dt = time.perf_counter() - t
print(f"depth {depth}: {jobs * len(qs) / dt / 1e6:.0f} M queries/s, {dt / jobs * 1e3:.2f} ms per batch")
pl.close()
