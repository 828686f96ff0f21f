"""Device encoder with parts skipped (debug build, ANX_ENC_DBG bits: 1 no count-vector writes, 2 no code stores, 4 no walk, 8 no
record stores; results are WRONG when set -- timing only): the [anx encode/device] laps of ANX_ENCODE_TIMING per variant.
usage: ANX_LIB=build/libanx_dbg.so enc_probe.py [nq]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ANX_ENCODE_TIMING"] = "1"
import analiticcl_amd as A
from analiticcl_amd import synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = synth.materialize_golden("/tmp/anxdata")
g = A.VariantModel(p["alphabet"], A.Weights(), device=0); g.read_lexicon(p["eng"]); g.build()
qs = synth.make_queries(synth.load_lexicon_words(p["eng"]), nq, max_len=16, seed=synth.SEED)
blob = b"".join(q.encode() + b"\0" for q in qs)
sp = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
for v in (0, 1, 2, 4, 8, 15, 0):
    os.environ["ANX_ENC_DBG"] = str(v)
    for rep in range(3):
        print(f"--- dbg={v} rep {rep}", file=sys.stderr, flush=True)
        b = g.encode_packed(blob, nq, sp)
        del b
