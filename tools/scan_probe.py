"""k_scan_bits / k_filter_score time of the bench workload with parts of the kernels skipped (ANX_SCAN_DBG bits: 1 one query
per pass, 2 no class tests, 4 signature walk only, 8 no hit expansion, 16 no band filter in the scan, 32 filter without its gathers, 64 nothing written; results are WRONG when set -- timing only), one
process, one encoded batch.  usage: scan_probe.py [eng|nld] [nq] [max_len] [d]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import analiticcl_amd as A
from analiticcl_amd import synth

lex = sys.argv[1] if len(sys.argv) > 1 else "eng"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
maxlen = int(sys.argv[3]) if len(sys.argv) > 3 else 16
d = int(sys.argv[4]) if len(sys.argv) > 4 else 2
p = synth.materialize_golden("/tmp/anxdata")
g = A.VariantModel(p["alphabet"], A.Weights(), device=0); g.read_lexicon(p[lex]); g.build()
qs = synth.make_queries(synth.load_lexicon_words(p[lex]), nq, max_len=maxlen, seed=synth.SEED)
b = g.encode_batch(qs, A.SearchParameters(max_anagram_distance=3, max_edit_distance=d, max_matches=10))
def run(label, env):
    for k in ("ANX_SCAN_DBG", "ANX_SCORE_DBG"): os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(2): b.run()
    acc = {}
    for _ in range(5):
        b.run(); st = b.stats()
        for k in ("ms_scan_kernel", "ms_filter_score_kernel", "ms_group", "ms_rank", "ms_total"): acc[k] = acc.get(k, 0.0) + st[k] / 5
    print(f"{label:28s} scan {acc['ms_scan_kernel']:.3f}  fscore {acc['ms_filter_score_kernel']:.3f}  compact {acc['ms_group']:.3f} rank {acc['ms_rank']:.3f} total {acc['ms_total']:.3f}  tiles {st['n_scan_blocks']} tests/q {st['n_class_tests']/nq:.0f} slots {st['n_pair_slots']} pairs {st['n_pairs']}", flush=True)
run("default", {})
for v in (8, 1, 2, 16, 32):
    run(f"scan dbg={v}", {"ANX_SCAN_DBG": str(v)})
for v in (1, 2):
    run(f"score dbg={v}", {"ANX_SCORE_DBG": str(v)})
run("default again", {})
