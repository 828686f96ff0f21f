"""Distribution of DL survivors per query (input of k_rank) on the bench workload sample."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import analiticcl_amd as A
from analiticcl_amd import synth
import sys as _s
LEX, MAXLEN, D = (_s.argv[1], int(_s.argv[2]), int(_s.argv[3])) if len(_s.argv) > 3 else ("eng", 16, 2)
d = synth.materialize_golden("/tmp/anxdata")
m = A.VariantModel(d["alphabet"], A.Weights(), device=0)
m.read_lexicon(d[LEX]); m.build()
qs = synth.make_queries(synth.load_lexicon_words(d[LEX]), 60000, max_len=MAXLEN)
p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=D, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
b = m.encode_batch(qs, p); b.run()
cnt = collections.Counter()
for (q, vid, ld, lcs, pre, suf, same, score) in b.fetch_pairs():
    if ld >= 0 and score >= 0.25: cnt[q] += 1
n = np.array([cnt.get(i, 0) for i in range(len(qs))])
print("mean", n.mean(), "max", n.max(), "sum n^2 / sum n", (n.astype(float)**2).sum() / n.sum())
for t in (0, 4, 8, 16, 32, 64, 128, 256, 512): print("<=", t, (n <= t).mean(), "share of n^2 above", ((n[n > t].astype(float))**2).sum() / (n.astype(float)**2).sum())
# rows the cutoff rule cannot drop (k_rank prunes the others before ranking): score > best / 2
best = collections.defaultdict(float)
pairs = [(q, score) for (q, vid, ld, lcs, pre, suf, same, score) in b.fetch_pairs() if ld >= 0 and score >= 0.25]
for q, s in pairs: best[q] = max(best[q], s)
kept = collections.Counter(q for q, s in pairs if not (s <= best[q] / 2.0 and s < best[q]))
k = np.array([kept.get(i, 0) for i in range(len(qs))])
print("kept after cutoff pruning: mean", k.mean(), "max", k.max(), "sum k^2 / n", (k.astype(float)**2).sum() / len(qs))
for t in (16, 32, 64, 128, 256): print("kept <=", t, (k <= t).mean(), "share of k^2 above", ((k[k > t].astype(float))**2).sum() / (k.astype(float)**2).sum())
