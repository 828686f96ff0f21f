"""Every share of the configs[3] job under the length-partitioned split, run one after the other on the one GPU: ms per share (the
8-GPU job takes the longest share), queries per tile, against a random eighth (consecutive input ranges).
usage: length_shares.py [job queries, default 10_000_000] [shards, default 8] [rounds, default 6]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import analiticcl_amd as A
from analiticcl_amd import synth

NJOB = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
paths = synth.materialize_golden("/tmp/anxdata")
words = list(dict.fromkeys(synth.load_lexicon_words(paths["eng"]) + synth.load_lexicon_words(paths["nld"])))
lex = synth.make_lexicon(words, 1_000_000, seed=11)
path = os.path.join(tempfile.gettempdir(), "anx_big.lexicon")
open(path, "w", encoding="utf-8").write("\n".join(lex) + "\n")
m = A.VariantModel(paths["alphabet"], A.Weights(), device=0); m.read_lexicon(path); m.build()
p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
t = time.time()
job = synth.make_queries(lex, NJOB, max_len=32, min_len=4, seed=6)
print(f"job of {NJOB} queries generated in {time.time() - t:.0f} s", flush=True)
lens = np.array([len(q.encode("utf-8")) for q in job], dtype=np.uint32)
def run(qs, label):
    b = m.encode_batch(qs, p)
    b.run(); b.run()
    t0 = time.perf_counter()
    for _ in range(3): b.run()
    dt = (time.perf_counter() - t0) / 3
    st = b.stats(); b.free()
    print(f"{label}: {len(qs)} queries {dt * 1e3:.2f} ms ({dt * 1e3 / (len(qs) / 1e6):.2f} per 1M) scan {st['ms_scan_kernel']:.2f} fs {st['ms_filter_score_kernel']:.2f} q/tile {len(qs) / max(st['n_scan_blocks'], 1):.1f} pairs/q {st['n_pairs'] / len(qs):.0f}", flush=True)
    return dt
dtr = run(job[: NJOB // S], "random share (consecutive range)")
# the split as successive calls of a multi-device model would see it: every round's share times correct the next round's cuts
for rnd in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
    gid = m.length_split(job, p, S)
    times = []
    for g in range(S):
        ix = np.nonzero(gid == g)[0]
        ls = lens[ix]
        times.append(run([job[i] for i in ix], f"round {rnd} share {g} (lengths {ls.min()}-{ls.max()})"))
    tot, worst = sum(times), max(times)
    print(f"round {rnd}: sum {tot * 1e3:.1f} ms, longest share {worst * 1e3:.2f} ms = the {S}-GPU job time ({NJOB / worst / 1e6:.0f} M queries/s); balance sum/{S}/longest {tot / S / worst:.2f}; "
          f"random shares {dtr * 1e3:.2f} ms each -> speed-up of the job {dtr / worst:.2f}x", flush=True)
    m.length_split(job, p, S, learn_ms=[t * 1e3 for t in times])
# per-length cost (ms per 1M queries) for the cost model
for L in range(4, 33, 4):
    ix = np.nonzero(lens == L)[0][:300_000]
    if ix.size > 50_000: run([job[i] for i in ix], f"length {L} only")
