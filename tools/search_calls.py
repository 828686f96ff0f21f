#!/usr/bin/env python3
"""anx_find_all_matches_batch on BASELINE configs[4]'s share (12.5 MB of text), N calls in a row: seconds per call, process CPU seconds, the cgroup's
throttled periods -- the spread between the calls of one process.  usage: search_calls.py [calls] [MB]"""
import ctypes as C
import os
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random

import analiticcl_amd as A
from analiticcl_amd import _lib as L
from analiticcl_amd import synth

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mb = float(sys.argv[2]) if len(sys.argv) > 2 else 12.5
paths = synth.materialize_golden(os.path.join(tempfile.gettempdir(), f"anx_bench_data_{os.getuid()}_0"))
words = synth.load_lexicon_words(paths["eng"])
m = A.VariantModel(paths["alphabet"], A.Weights(), device=0)
m.read_lexicon(paths["eng"])
rng = random.Random(7)
common = [w for w in words if w.isalpha()][::23][:5000]
LM = A.VocabParams(vocabtype="LM")
for _ in range(20000):
    m.add_to_vocabulary(f"{rng.choice(common)} {rng.choice(common)}", rng.randrange(1, 20), LM)
for w in common[:500]:
    m.add_to_vocabulary(f"<bos> {w}", 5, LM)
m.build()
texts = synth.make_running_text(common, mb, seed=7)
nbytes = sum(len(t.encode("utf-8")) for t in texts)
sp = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, max_ngram=3, score_threshold=0.25, cutoff_threshold=2.0)
arr = (C.c_char_p * len(texts))(*[t.encode("utf-8") for t in texts])
spc = sp._c_search()


def throttled():
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(p):
                if line.startswith("nr_throttled"):
                    return int(line.split()[1])
        except OSError:
            pass
    return 0


out = []
for i in range(calls):
    ms, offs, rows, nrows = C.POINTER(L.Match)(), C.POINTER(C.c_size_t)(), C.POINTER(L.Result)(), C.c_size_t(0)
    t, c0, th0 = time.perf_counter(), sum(os.times()[:2]), throttled()
    L.check(L.lib().anx_find_all_matches_batch(m.h, arr, len(texts), C.byref(spc), C.byref(ms), C.byref(offs), C.byref(rows), C.byref(nrows), None))
    dt = time.perf_counter() - t
    L.lib().anx_matches_free(ms, offs, rows, None)
    out.append((round(dt * 1e3, 1), round(sum(os.times()[:2]) - c0, 2), throttled() - th0))
print("ms per call, process CPU s, throttled periods:", out)
s = sorted(x[0] for x in out[4:])
print(f"{nbytes / 1e6:.1f} MB; calls 5..: best {nbytes / 1e3 / s[0]:.0f} MB/s, median {nbytes / 1e3 / s[len(s) // 2]:.0f} MB/s, worst {nbytes / 1e3 / s[-1]:.0f} MB/s")
