"""A fixed-seed slice of the randomised soaks inside `pytest -m gpu` (the open-ended runs are tools/fuzz_parity.py,
tools/fuzz_search.py, tools/fuzz_confusables_device.py on the GPU box): random sub-lexicons with and without frequencies and
variant lists, random score weights, thresholds, max_matches, cutoffs, StopAtExactMatch and freq_weight against the C oracle
(ranked ids, f64 scores compared with ==, per-query scored-pair counts); random texts through search mode against the twin (every
Match field; the lattices are decoded on the device unless the round has context rules)."""
import time

import pytest

pytestmark = pytest.mark.gpu

import soak_common as S


def test_parity_soak_fixed_seeds():
    t0 = time.time()
    done = 0
    for seed in range(7000, 7040):
        S.parity_round(seed, max_words=8000)
        done += 1
        if time.time() - t0 > 45 and done >= 12:   # a slow box still checks a dozen configurations
            break
    assert done >= 12


def test_search_soak_fixed_seeds():
    t0 = time.time()
    worlds = {}
    done = 0
    for seed in range(9000, 9040):
        S.search_round(seed, worlds)
        done += 1
        if time.time() - t0 > 45 and done >= 12:
            break
    assert done >= 12
