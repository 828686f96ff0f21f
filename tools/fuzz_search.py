"""Randomised search-mode soak on the GPU: the small world of tests/test_gpu_search.py (3000 words + phrases, optional bigram LM
and context rules), fresh random texts per round (short stretches, long stretches, varying max_seq) -- every Match field of
anx_find_all_matches_batch against the Python twin (test infrastructure).  usage: fuzz_search.py [seconds] [first seed]"""
import os, random, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_search as T
from analiticcl_amd import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
rounds = 0
worlds = {}
from soak_common import search_round   # one round: shared with tests/test_gpu_soak.py
while time.time() - t0 < budget:
    with_lm, with_rules, max_seq, ntexts, n_multi, n_tagged = search_round(seed, worlds)
    rounds += 1
    print(f"seed {seed}: ok  lm {with_lm} rules {with_rules} max_seq {max_seq} texts {ntexts} n-gram matches {n_multi} tagged {n_tagged}", flush=True)
    seed += 1
print(f"{rounds} rounds identical to the twin in {time.time() - t0:.0f} s")
