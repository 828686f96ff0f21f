"""Randomised parity soak on the GPU: random sub-lexicons (with and without frequencies, sometimes with variant lists), random
score weights and search parameters, random queries (perturbed words, raw noise, long strings) -- ranked ids, f64 scores and
per-query scored-pair counts of the product against the C oracle (test infrastructure).  Stops at the first difference with the
seed that reproduces it.  usage: fuzz_parity.py [seconds] [first seed]"""
import os, random, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import cwrap as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
from soak_common import parity_round as one   # one round: shared with tests/test_gpu_soak.py


t0 = time.time()
seed = seed0
done = 0
while time.time() - t0 < budget:
    info = one(seed)
    done += 1
    print(f"seed {seed}: ok  words {info[0]} freq {info[1]} variants {info[2]} queries {info[3]} pairs {info[4]} params {info[5]}", flush=True)
    seed += 1
print(f"{done} configurations identical to the oracle in {time.time() - t0:.0f} s")
