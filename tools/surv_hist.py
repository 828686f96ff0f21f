"""Distribution of DL survivors per query (input of k_rank) on the bench workload sample."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import analiticcl_amd as A
from analiticcl_amd import synth
d = synth.materialize_golden("/tmp/anxdata")
m = A.VariantModel(d["alphabet"], A.Weights(), device=0)
m.read_lexicon(d["eng"]); m.build()
qs = synth.make_queries(synth.load_lexicon_words(d["eng"]), 60000, max_len=16)
p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
b = m.encode_batch(qs, p); b.run()
cnt = collections.Counter()
for (q, vid, ld, lcs, pre, suf, same, score) in b.fetch_pairs():
    if ld >= 0 and score >= 0.25: cnt[q] += 1
n = np.array([cnt.get(i, 0) for i in range(len(qs))])
print("mean", n.mean(), "max", n.max(), "sum n^2 / sum n", (n.astype(float)**2).sum() / n.sum())
for t in (0, 4, 8, 16, 32, 64, 128, 256, 512): print("<=", t, (n <= t).mean(), "share of n^2 above", ((n[n > t].astype(float))**2).sum() / (n.astype(float)**2).sum())
