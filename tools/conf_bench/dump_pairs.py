"""(query, ranked vocab ids) of BASELINE configs[2]-shaped queries, from the C oracle (test infrastructure), for
tools/conf_bench/main.cpp: one line per query `query<TAB>id id id ...`.  usage: dump_pairs.py <out.tsv> [queries]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from analiticcl_amd import synth
from oracle import cwrap as O

out, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3000
d = synth.materialize_golden("/tmp/anxdata")
o = O.OracleModel(alphabet_path=d["alphabet"])
o.read_lexicon(d["nld"])
o.build()
words = synth.load_lexicon_words(d["nld"])
qs = synth.make_queries(words, n, max_len=24, seed=synth.SEED + 2)
p = O.make_params(("abs", 3), ("abs", 3), 10, 0.25, 0.0)  # the cutoff follows the rescoring: rows up to the crop
with open(out, "w") as f:
    for q in qs:
        if "\t" in q or "\n" in q:
            continue
        rows = o.find_variants(q, p)
        f.write(q + "\t" + " ".join(str(r[0]) for r in rows) + "\n")
