import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from analiticcl_amd import synth
from oracle import cwrap as O
d = synth.materialize_golden("/tmp/anxdata")
o = O.OracleModel(alphabet_path=d["alphabet"]); o.read_lexicon(d["eng"]); o.build()
qs = synth.make_queries(synth.load_lexicon_words(d["eng"]), 40000, max_len=16)
op = O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0)
for nt in (1, 8, 16, 32, 64, 128, 256):
    n = 400 if nt == 1 else min(40000, 1500 * nt)
    t = time.time(); rc, _r, _c, tp, _tc = o.find_variants_batch(qs[:n], op, nthreads=nt, stride=16); dt = time.time() - t
    print(nt, "threads:", round(n / dt), "q/s", round(tp / dt), "pairs/s")
