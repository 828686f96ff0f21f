"""BASELINE.json configs[3], one GPU's share: merged 1 M-entry synthetic lexicon (eng.aspell + nld.aspell + Markov-chain
words, len 4-32), 1.25 M length-bucketed queries len 4-32, k=3 d=2 n=10.  Prints build / encode / run times, the batch
statistics and a 150-query spot check against the C oracle (skipped with a third argument "nocheck": timing of the whole
10 M-query job on one GPU).  usage: big_lexicon_bench.py [entries] [queries] [nocheck]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import cwrap as O

NE = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 1_250_000
d = synth.materialize_golden("/tmp/anxdata")
words = list(dict.fromkeys(synth.load_lexicon_words(d["eng"]) + synth.load_lexicon_words(d["nld"])))
t = time.time(); lex = synth.make_lexicon(words, NE, seed=11); print("lexicon: %d entries, %.1f s" % (len(lex), time.time() - t))
path = "/tmp/anx_big.lexicon"
open(path, "w", encoding="utf-8").write("\n".join(lex) + "\n")
t = time.time()
g = A.VariantModel(d["alphabet"], A.Weights(), device=0); g.read_lexicon(path); g.build()
print("GPU model build: %.1f s, %d classes" % (time.time() - t, g.num_classes()))
qs = synth.make_queries(lex, NQ, max_len=32, min_len=4, seed=5)
qs.sort(key=len)  # length-bucketed
p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
t = time.time(); b = g.encode_batch(qs, p); te = time.time() - t
for _ in range(2): b.run()
t = time.time()
for _ in range(5): b.run()
tr = (time.time() - t) / 5
st = b.stats()
print("encode %.2f s; run %.2f ms per batch of %d queries = %.2f ms per 1M; %.1f G pairs/s; pairs/query %.1f, class tests/query %.0f, slots %d, survivors %d, results %d"
      % (te, tr * 1e3, NQ, tr * 1e3 * 1e6 / NQ, st["n_pairs"] / tr / 1e9, st["n_pairs"] / NQ, st["n_class_tests"] / NQ, st["n_pair_slots"], st["n_survivors"], st["n_results"]))
print({k: round(st[k], 3) for k in ("ms_scan", "ms_score", "ms_group", "ms_rank", "ms_total", "ms_scan_kernel", "ms_filter_score_kernel")}, "tiles", st["n_scan_blocks"], "adj tiles", st["n_adj_tiles"], "adj records", st["n_adj_records"], "first", st["n_adj_records_first"])
if len(sys.argv) > 3 and sys.argv[3] == "nocheck":
    sys.exit(0)
res = b.fetch()
t = time.time()
o = O.OracleModel(alphabet_path=d["alphabet"]); o.read_lexicon(path); o.build()
print("oracle build %.1f s" % (time.time() - t))
op = O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0, False, 0.0)
import random
idx = random.Random(1).sample(range(NQ), 150)
bad = 0
for i in idx:
    exp = o.find_variants(qs[i], op)
    if [tuple(x) for x in res[i]] != exp:
        bad += 1
        if bad <= 3: print("MISMATCH", repr(qs[i]), res[i][:3], exp[:3])
print("spot check: %d mismatches / %d" % (bad, len(idx)))
sys.exit(1 if bad else 0)
