"""The device memory pool and the pinned result cache under pressure: with caches of a few megabytes (ANX_POOL_CACHE_MB,
ANX_PINNED_CACHE_MB; read once per process, hence the child process) every batch of a changing mix of sizes evicts the oldest cached
blocks and allocates afresh -- same rows as with the default caches, for the staged calls and a search-mode call in parts."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import synth
from fullsize_common import checksum

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import analiticcl_amd as A
from analiticcl_amd import synth
from fullsize_common import checksum
import test_gpu_caches as T
print("RESULT", *T.workload(sys.argv[2]))
"""


def workload(data_dir):
    lex = os.path.join(data_dir, "eng.aspell.lexicon")
    words = synth.load_lexicon_words(lex)
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(lex)
    g.build()
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    out = []
    for rnd, n in enumerate((3_000, 120_000, 900, 60_000, 250_000, 5_000, 120_000)):
        qs = synth.make_queries(words, n, max_len=16, seed=100 + rnd)
        b = g.encode_batch(qs, p)
        b.run()
        out.append(checksum(*b.fetch_arrays()))
        b.free()
    common = [w for w in words if w.isalpha()][::23][:3000]
    texts = synth.make_running_text(common, 0.7, seed=9)
    A.set_switch("ANX_SEARCH_PARTS_MIN", "1")
    A.set_switch("ANX_SEARCH_PARTS", "3")
    try:
        for _ in range(2):
            off, ma, ra = g.find_all_matches_arrays(texts, A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, max_ngram=2))
            out.append(int(off[-1]) * 1_000_003 + int(ra.shape[0]))
            out.append(int(np.bitwise_xor.reduce(ra["vocab_id"].astype(np.uint64) * (np.arange(ra.shape[0], dtype=np.uint64) | np.uint64(1)))) if ra.shape[0] else 0)
    finally:
        A.set_switch("ANX_SEARCH_PARTS", None)
        A.set_switch("ANX_SEARCH_PARTS_MIN", None)
    return out


def test_small_caches_give_the_same_rows(data_dir):
    ref = workload(data_dir)
    env = dict(os.environ, ANX_POOL_CACHE_MB="48", ANX_PINNED_CACHE_MB="6")
    r = subprocess.run([sys.executable, "-c", CHILD, REPO, data_dir], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
    got = [int(x) for x in line.split()[1:]]
    assert got == ref
