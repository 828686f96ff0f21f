"""Device-vs-host soak of the confusable weighting (GPU box): random pattern sets over the golden lexicons, random parameters,
late and early mode; every ranked row (ids, order, f64 scores) of the device path (conf.hip) must equal the host path
(ANX_CONFUSABLES=host).  One round = tests/soak_common.py conf_round (tests/test_gpu_soak.py runs a fixed-seed slice of them).
usage: fuzz_confusables_device.py [seconds] [seed]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import soak_common as S

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
rows = rounds = 0
while time.time() - t0 < budget:
    rows += S.conf_round(seed * 100003 + rounds)
    rounds += 1
print(f"{rounds} rounds, {rows} ranked rows: device == host in {time.time() - t0:.0f} s")
