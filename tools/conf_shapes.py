"""Shapes of the (input, candidate) pairs the confusable weighting sees on BASELINE configs[2]: lengths of the two middles once the
common prefix and suffix are gone (what decides the route through the edit script).  usage: conf_shapes.py [queries]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import analiticcl_amd as A
from analiticcl_amd import synth
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
p = synth.materialize_golden("/tmp/anxdata")
words = synth.load_lexicon_words(p["nld"])
qs = synth.make_queries(words, nq, max_len=24, seed=synth.SEED + 2)
m = A.VariantModel(p["alphabet"], A.Weights(), device=0); m.read_lexicon(p["nld"]); m.build()
sp = A.SearchParameters(max_anagram_distance=3, max_edit_distance=3, max_matches=10)
res = m.find_variants_ids(qs, sp)
h = collections.Counter(); n = 0
for q, rows in zip(qs, res):
    for (v, d, f) in rows:
        c = m.vocab_text(v)
        pfx = 0
        while pfx < min(len(q), len(c)) and q[pfx] == c[pfx]: pfx += 1
        sfx = 0
        while sfx < min(len(q), len(c)) - pfx and q[-1 - sfx] == c[-1 - sfx]: sfx += 1
        ma, mb = len(q) - pfx - sfx, len(c) - pfx - sfx
        h[(min(ma, 4), min(mb, 4))] += 1; n += 1
print("rows", n)
for k, v in sorted(h.items(), key=lambda kv: -kv[1])[:16]: print(k, f"{100.0 * v / n:.1f} %")
