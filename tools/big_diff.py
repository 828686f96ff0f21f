"""One-off large differential: GPU engine vs the C oracle on N random queries for several parameter sets
(ranked vocab ids identical, scores identical).  usage: big_diff.py [N] [lexicon eng|nld]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import cwrap as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
LEX = sys.argv[2] if len(sys.argv) > 2 else "eng"
d = synth.materialize_golden("/tmp/anxdata")
g = A.VariantModel(d["alphabet"], A.Weights(), device=0); g.read_lexicon(d[LEX]); g.build()
o = O.OracleModel(alphabet_path=d["alphabet"]); o.read_lexicon(d[LEX]); o.build()
qs = synth.make_queries(synth.load_lexicon_words(d[LEX]), N, max_len=28, seed=99)
sets = [dict(k=3, d=2, n=10, thr=0.25, cut=2.0, stop=False, fw=0.0), dict(k=3, d=3, n=20, thr=0.25, cut=2.0, stop=False, fw=0.0),
        dict(k=3, d=2, n=10, thr=0.25, cut=2.0, stop=True, fw=0.0), dict(k=2, d=2, n=3, thr=0.1, cut=1.2, stop=False, fw=0.3),
        dict(k=4, d=4, n=0, thr=0.4, cut=0.0, stop=False, fw=0.0)]
bad = 0
for ps in sets:
    gp = A.SearchParameters(max_anagram_distance=ps["k"], max_edit_distance=ps["d"], max_matches=ps["n"], score_threshold=ps["thr"],
                            cutoff_threshold=ps["cut"], stop_criterion=ps["stop"], freq_weight=ps["fw"])
    op = O.make_params(("abs", ps["k"]), ("abs", ps["d"]), ps["n"], ps["thr"], ps["cut"], ps["stop"], ps["fw"])
    t = time.time(); got = g.find_variants_ids(qs, gp); tg = time.time() - t
    t = time.time()
    stride = 4096 if ps["n"] == 0 else 64
    rc, res, counts, _tp, _tc = o.find_variants_batch(qs, op, nthreads=0, stride=stride)
    to = time.time() - t
    nbad = 0
    for i, q in enumerate(qs):
        exp = [(res[i * stride + j].vocab_id, res[i * stride + j].dist_score, res[i * stride + j].freq_score) for j in range(min(counts[i], stride))]
        if counts[i] > stride or counts[i] <= 0:  # the batch entry reports lists longer than `stride` as 0 (rc -1)
            exp = o.find_variants(q, op)
        if [tuple(x) for x in got[i]] != exp:
            nbad += 1
            if nbad <= 3: print("MISMATCH", ps, repr(q), got[i][:4], exp[:4])
    bad += nbad
    print(ps, "gpu %.2fs oracle %.2fs mismatches %d / %d" % (tg, to, nbad, len(qs)))
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
