"""Search-mode throughput (BASELINE config 5 shape, scaled): sentences of 5-25 sampled + perturbed lexicon words,
max_ngram 3, bigram LM counts from the same sampler.  usage: search_bench.py [MB of text] [parts,parts,...]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import analiticcl_amd as A
from analiticcl_amd import synth

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
d = synth.materialize_golden("/tmp/anxdata")
words = synth.load_lexicon_words(d["eng"])
rng = random.Random(7)
m = A.VariantModel(d["alphabet"], A.Weights(), device=0)
m.read_lexicon(d["eng"])
LM = A.VocabParams(vocabtype="LM")
common = [w for w in words if w.isalpha()][::23][:5000]
for _ in range(20000):
    a, b = rng.choice(common), rng.choice(common)
    m.add_to_vocabulary(f"{a} {b}", rng.randrange(1, 20), LM)
for w in common[:500]:
    m.add_to_vocabulary(f"<bos> {w}", 5, LM)
m.build()
pert = synth.make_queries(common, int(mb * 1e6 / 7) + 100, max_len=16, seed=3)
texts, cur, size, k = [], [], 0, 0
while size < mb * 1e6:
    n = rng.randrange(5, 26)
    sent = " ".join(pert[k:k + n]) + rng.choice([". ", "\n", ", ", "\n\n"])
    k += n
    cur.append(sent)
    if len(cur) == 8:
        texts.append("".join(cur)); size += len(texts[-1]); cur = []
p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, max_ngram=3,
                       lm_weight=float(os.environ.get("ANX_BENCH_LM_WEIGHT", "1.0")), max_seq=int(os.environ.get("ANX_BENCH_MAX_SEQ", "250")))
import ctypes as C
from analiticcl_amd import _lib as L
arr = (C.c_char_p * len(texts))(*[t.encode() for t in texts])
spc = p._c_search()
def _throttled():  # CFS quota: periods in which the cgroup ran out of CPU time
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                if line.startswith("nr_throttled"): return int(line.split()[1])
        except OSError:
            pass
    return 0
for rep in range(4):  # the C entry point alone (what a Rust / C caller sees)
    ms, offs, rows, nrows = C.POINTER(L.Match)(), C.POINTER(C.c_size_t)(), C.POINTER(L.Result)(), C.c_size_t(0)
    t = time.time(); c0 = sum(os.times()[:2]); th0 = _throttled()
    L.check(L.lib().anx_find_all_matches_batch(m.h, arr, len(texts), C.byref(spc), C.byref(ms), C.byref(offs), C.byref(rows), C.byref(nrows), None))
    dt = time.time() - t
    print(f"C ABI: {size/1e6:.1f} MB in {dt:.2f} s = {size/1e6/dt:.2f} MB/s, {offs[len(texts)]} matches, {nrows.value} variant rows; "
          f"process CPU {sum(os.times()[:2]) - c0:.2f} core-s, cgroup throttled periods +{_throttled() - th0}")
    L.lib().anx_matches_free(ms, offs, rows, None)
if len(sys.argv) > 2:  # search_bench.py MB parts[:MB per part],...: ANX_SEARCH_PARTS / _PART_BYTES settings side by side, alternating, 4 calls each
    settings = sys.argv[2].split(",")
    times = {k: [] for k in settings}
    for rep in range(4):
        for k in settings:
            A.set_switch("ANX_SEARCH_PARTS", k.split(":")[0])
            A.set_switch("ANX_SEARCH_PART_BYTES", str(int(float(k.split(":")[1]) * (1 << 20))) if ":" in k else None)
            ms, offs, rows, nrows = C.POINTER(L.Match)(), C.POINTER(C.c_size_t)(), C.POINTER(L.Result)(), C.c_size_t(0)
            t = time.time()
            L.check(L.lib().anx_find_all_matches_batch(m.h, arr, len(texts), C.byref(spc), C.byref(ms), C.byref(offs), C.byref(rows), C.byref(nrows), None))
            times[k].append(time.time() - t)
            L.lib().anx_matches_free(ms, offs, rows, None)
    for k in settings:
        ts = sorted(times[k])
        print(f"parts {k}: best {size/1e6/ts[0]:.1f} MB/s, median {size/1e6/ts[len(ts)//2]:.1f} MB/s")
    sys.exit(0)
for rep in range(1):
    t = time.time()
    res = m.find_all_matches_ids(texts, p)
    dt = time.time() - t
    nm = sum(len(r) for r in res)
    print(f"{size/1e6:.1f} MB in {len(texts)} texts: {dt:.2f} s = {size/1e6/dt:.2f} MB/s, {nm} matches, {nm/dt:.0f} matches/s")
