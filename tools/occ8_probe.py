"""Experiment: runs one (d, max_len) batch on the occupancy-forced build build/libanx_occ8.so
(tools/build_occ8_probe.sh) with the inline-wide kernel variant (ANX_FS_SPLIT=0), checks 300 queries against the C oracle and
prints the kernel time.  Run under `timeout`: a hang shows up as exit code 124.  usage: occ8_probe.py d max_len [nq]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault("ANX_FS_SPLIT", "0")
from analiticcl_amd import _lib
if os.environ.get("ANX_PROBE_LIB", "occ8") == "occ8":
    _lib.LIB_PATH = os.path.join(R, "build", "libanx_occ8.so")
import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import cwrap as O

d, maxlen = int(sys.argv[1]), int(sys.argv[2])
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
p = synth.materialize_golden("/tmp/anxdata")
g = A.VariantModel(p["alphabet"], A.Weights(), device=0); g.read_lexicon(p["eng"]); g.build()
qs = synth.make_queries(synth.load_lexicon_words(p["eng"]), nq, max_len=maxlen, seed=17)
gp = A.SearchParameters(max_anagram_distance=3, max_edit_distance=d, max_matches=10)
b = g.encode_batch(qs, gp)
for _ in range(3): b.run()
st = b.stats()
print(f"d={d} max_len={maxlen} nq={nq}: filter_score {st['ms_filter_score_kernel']:.3f} ms, total {st['ms_total']:.3f} ms, pairs {st['n_pairs']}", flush=True)
res = b.fetch()
o = O.OracleModel(alphabet_path=p["alphabet"]); o.read_lexicon(p["eng"]); o.build()
op = O.make_params(("abs", 3), ("abs", d), 10, 0.25, 2.0)
bad = sum([tuple(x) for x in res[i]] != o.find_variants(qs[i], op) for i in range(0, nq, max(1, nq // 300)))
print("mismatches", bad, flush=True)
sys.exit(1 if bad else 0)
