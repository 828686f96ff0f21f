"""A fixed-seed slice of the randomised soaks inside `pytest -m gpu` (the open-ended runs are tools/fuzz_parity.py,
tools/fuzz_search.py, tools/fuzz_confusables_device.py on the GPU box): random sub-lexicons with and without frequencies and
variant lists, random score weights, thresholds, max_matches, cutoffs, StopAtExactMatch and freq_weight against the C oracle
(ranked ids, f64 scores compared with ==, per-query scored-pair counts); random texts through search mode against the twin (every
Match field; the lattices are decoded on the device unless the round has context rules); random confusable pattern sets, device
weighting against the host path, every ranked row."""
import time

import pytest

pytestmark = pytest.mark.gpu

import soak_common as S


def test_parity_soak_fixed_seeds():
    """150 random configurations (~0.7 s each on the pool's boxes); a box several times slower still has to get through 60."""
    t0 = time.time()
    done = 0
    for seed in range(7000, 7150):
        S.parity_round(seed, max_words=8000)
        done += 1
        if time.time() - t0 > 330 and done >= 60:
            break
    assert done >= 60
    print(f"parity soak: {done} configurations in {time.time() - t0:.0f} s")


def test_search_soak_fixed_seeds():
    """150 rounds of random texts through search mode against the twin (~1.3 s each)."""
    t0 = time.time()
    worlds = {}
    done = 0
    for seed in range(9000, 9150):
        S.search_round(seed, worlds)
        done += 1
        if time.time() - t0 > 420 and done >= 60:
            break
    assert done >= 60
    print(f"search soak: {done} rounds in {time.time() - t0:.0f} s")


def test_confusable_soak_fixed_seeds():
    """>= 20 M ranked rows: the device-side confusable weighting against the host path, random pattern sets and parameters."""
    t0 = time.time()
    rows = rounds = 0
    for seed in range(11000, 11200):
        rows += S.conf_round(seed)
        rounds += 1
        if rows >= 20_000_000 or (time.time() - t0 > 300 and rows >= 5_000_000):
            break
    assert rows >= 5_000_000
    print(f"confusable soak: {rounds} rounds, {rows} ranked rows in {time.time() - t0:.0f} s")
