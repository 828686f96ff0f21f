"""Device time of BASELINE configs[2] (nld.aspell, 1 M queries len <= 24, d = 3, 10 confusable patterns, weighted on the device):
ms per pass over the resident batch, with and without the confusable weighting (ANX_CONFUSABLES=off is not a product mode: the
second model simply has no patterns).  For per-kernel times run it under tools/trace_cmd.sh.  usage: conf_probe.py [nq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import analiticcl_amd as A
from analiticcl_amd import synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = synth.materialize_golden("/tmp/anxdata")
words = synth.load_lexicon_words(p["nld"])
qs = synth.make_queries(words, nq, max_len=24, seed=synth.SEED + 2)
sp = A.SearchParameters(max_anagram_distance=3, max_edit_distance=3, max_matches=10)
for label, conf in (("with confusables", True), ("without", False)):
    m = A.VariantModel(p["alphabet"], A.Weights(), device=0)
    m.read_lexicon(p["nld"])
    if conf:
        m.read_confusablelist(os.path.join(synth.GOLDEN_DATA, "confusables10.tsv"))
    m.build()
    b = m.encode_batch(qs, sp)
    for _ in range(2):
        b.run()
    t0 = time.perf_counter()
    for _ in range(5):
        b.run()
    dt = (time.perf_counter() - t0) / 5
    print(f"{label:18s} {dt * 1e3:8.2f} ms per pass, {b.stats()['n_results']} rows", flush=True)
    b.free()
