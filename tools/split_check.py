"""First-call balance of the length-partitioned split on a job other than configs[3]'s (the prior's constants were fitted there):
a golden lexicon (eng | nld), N queries of up to max_len symbols, S shares run one after the other on one GPU.
usage: split_check.py [eng|nld] [queries, default 4_000_000] [max_len, default 16] [d, default 2] [shares, default 8] [rounds, default 2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import analiticcl_amd as A
from analiticcl_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "eng"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
ML = int(sys.argv[3]) if len(sys.argv) > 3 else 16
D = int(sys.argv[4]) if len(sys.argv) > 4 else 2
S = int(sys.argv[5]) if len(sys.argv) > 5 else 8
R = int(sys.argv[6]) if len(sys.argv) > 6 else 2
paths = synth.materialize_golden("/tmp/anxdata")
m = A.VariantModel(paths["alphabet"], A.Weights(), device=0); m.read_lexicon(paths[which]); m.build()
p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=D, max_matches=10)
job = synth.make_queries(synth.load_lexicon_words(paths[which]), N, max_len=ML, seed=77)
def run(qs):
    b = m.encode_batch(qs, p)
    b.run(); b.run()
    t0 = time.perf_counter()
    for _ in range(3): b.run()
    dt = (time.perf_counter() - t0) / 3
    b.free()
    return dt
for rnd in range(R):
    gid = m.length_split(job, p, S)
    times = [run([job[i] for i in np.nonzero(gid == g)[0]]) for g in range(S)]
    print(f"{which} len<={ML} d={D} round {rnd}: shares (ms) {[round(t * 1e3, 2) for t in times]} balance {sum(times) / S / max(times):.2f}", flush=True)
    m.length_split(job, p, S, learn_ms=[t * 1e3 for t in times])
