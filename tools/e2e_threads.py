"""Where do concurrent callers wait?  N host threads each run encode_packed -> run (own stream) -> fetch_arrays -> free on
the same model; per phase the mean wall time per thread is printed next to the single-thread figure, plus the aggregate
rate.  usage: e2e_threads.py [threads ...]   (default 1 2 3)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import analiticcl_amd as A
from analiticcl_amd import synth

d = synth.materialize_golden("/tmp/anxdata")
m = A.VariantModel(d["alphabet"], A.Weights(), device=0)
m.read_lexicon(d["eng"])
m.build()
N = 1000000
qs = synth.make_queries(synth.load_lexicon_words(d["eng"]), N, max_len=16)
p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
packed = ("\0".join(qs) + "\0").encode("utf-8")
PER = 6


def worker(out, st):
    for _ in range(PER):
        t0 = time.perf_counter()
        b = m.encode_packed(packed, N, p)
        t1 = time.perf_counter()
        b.run(st.cuda_stream)
        t2 = time.perf_counter()
        r = b.fetch_arrays()
        t3 = time.perf_counter()
        del r
        b.free()
        t4 = time.perf_counter()
        out.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))


for nthr in [int(x) for x in sys.argv[1:]] or [1, 2, 3]:
    streams = [torch.cuda.Stream() for _ in range(nthr)]
    th = [threading.Thread(target=worker, args=([], st)) for st in streams]  # warm the buffer pools of nthr concurrent batches
    for x in th:
        x.start()
    for x in th:
        x.join()
    outs = [[] for _ in range(nthr)]
    th = [threading.Thread(target=worker, args=(o, st)) for o, st in zip(outs, streams)]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    wall = time.perf_counter() - t
    rows = [r for o in outs for r in o]
    mean = [sum(r[i] for r in rows) / len(rows) * 1e3 for i in range(4)]
    print("%d thread(s): encode %.2f ms  run %.2f ms  fetch %.2f ms  free %.2f ms per batch and thread -> %.1f M queries/s aggregate"
          % (nthr, mean[0], mean[1], mean[2], mean[3], nthr * PER * N / wall / 1e6), flush=True)
