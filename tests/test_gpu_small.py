"""The small call (analiticcl_amd/csrc/small_path.hpp): anx_find_variants_batch for a few inputs -- the reference's own granularity, one
string per call (src/lib.rs:972), 1 000 per batch (src/bin/analiticcl.rs:416) -- against the batch pipeline (ANX_SMALL=0) and the oracle."""
import ctypes as C
import os
import random
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import _lib as L
from analiticcl_amd import synth
from oracle import cwrap as O


def call(model, qs, params):
    """anx_find_variants_batch(char**) -> [[(vocab_id, dist, freq)]]"""
    lib = L.lib()
    arr = (C.c_char_p * len(qs))(*[q.encode("utf-8") if isinstance(q, str) else q for q in qs])
    rows = C.POINTER(L.Result)()
    offs = C.POINTER(C.c_size_t)()
    cp = params._c()
    L.check(lib.anx_find_variants_batch(model.h, arr, len(qs), C.byref(cp), C.byref(rows), C.byref(offs)))
    try:
        off = [offs[i] for i in range(len(qs) + 1)]
        return [[(rows[j].vocab_id, rows[j].dist_score, rows[j].freq_score) for j in range(off[i], off[i + 1])] for i in range(len(qs))]
    finally:
        lib.anx_results_free(rows, offs)


def small_stats():
    out = (C.c_uint64 * 2)()
    assert L.lib().anx_debug_small_stats(out) == 0
    return out[0], out[1]


@pytest.fixture(scope="module")
def eng(data_dir):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    g.build()
    o = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
    o.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    o.build()
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    return g, o, words


def via_batch_path(model, qs, params):
    A.set_switch("ANX_SMALL", "0")
    try:
        return call(model, qs, params)
    finally:
        A.set_switch("ANX_SMALL", None)


@pytest.mark.parametrize("n", [1, 2, 7, 64, 1000, 4096])
def test_small_call_equals_batch_path_and_oracle(eng, n):
    g, o, words = eng
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
    op = O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0)
    qs = synth.make_queries(words, n, max_len=16, seed=500 + n)
    t0 = small_stats()
    got = call(g, qs, p)
    t1 = small_stats()
    assert t1[0] == t0[0] + 1, "the small path did not take the call"
    assert got == via_batch_path(g, qs, p)
    assert small_stats()[0] == t1[0]       # (the A/B call did not)
    for i in random.Random(n).sample(range(n), min(n, 300)):
        assert got[i] == o.find_variants(qs[i], op), qs[i]


@pytest.mark.parametrize("kw", [
    dict(max_anagram_distance=3, max_edit_distance=3, max_matches=20),
    dict(max_anagram_distance=2, max_edit_distance=1, max_matches=5),
    dict(max_anagram_distance=0.4, max_edit_distance=0.3, max_matches=10),
    dict(max_anagram_distance=(0.5, 3), max_edit_distance=(0.4, 2), max_matches=0),
    dict(max_anagram_distance=3, max_edit_distance=2, max_matches=3, score_threshold=0.6, cutoff_threshold=1.2),
    dict(max_anagram_distance=3, max_edit_distance=2, max_matches=10, freq_weight=0.5),
    dict(max_anagram_distance=5, max_edit_distance=4, max_matches=10),
])
def test_parameter_sets(eng, kw):
    g, _o, words = eng
    p = A.SearchParameters(**kw)
    qs = synth.make_queries(words, 700, max_len=24, seed=77) + ["", "a", "I", "zzzzzzzzzzzzzzzz", "héllo wörld", "日本語", "x" * 64, "separate!"]
    t0 = small_stats()
    got = call(g, qs, p)
    assert sum(small_stats()) == sum(t0) + 1
    assert got == via_batch_path(g, qs, p)


def test_inputs_the_small_path_hands_over(eng):
    """An input beyond 64 bytes, a call beyond 4096 inputs, StopAtExactMatch: the batch pipeline answers, the results are the same."""
    g, o, words = eng
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    op = O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0)
    t0 = small_stats()
    qs = ["recieve", "x" * 65, "seperate"]
    got = call(g, qs, p)
    assert small_stats() == t0
    assert got[0] == o.find_variants("recieve", op) and got[2] == o.find_variants("seperate", op) and got[1] == []
    qs = synth.make_queries(words, 4097, max_len=16, seed=3)
    got = call(g, qs, p)
    assert small_stats() == t0 and got[17] == o.find_variants(qs[17], op)
    ps = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, stop_criterion=True)
    got = call(g, ["separate", "seperate"], ps)
    assert small_stats() == t0
    assert got == [o.find_variants(q, O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0, True)) for q in ("separate", "seperate")]


def test_capacity_overflow_falls_back(eng):
    """Thousands of two- and three-letter queries: hundreds of scored pairs each, far beyond what the fixed buffers of the small path are
    sized for on average.  Either they fit or the call is redone by the batch pipeline -- the rows are the same."""
    g, _o, words = eng
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=3, max_matches=0, score_threshold=0.0, cutoff_threshold=0.0)
    short = sorted({w for w in words if 3 <= len(w) <= 4 and w.isalpha()})[:4096]
    assert len(short) == 4096
    got = call(g, short, p)
    assert got == via_batch_path(g, short, p)
    assert sum(len(r) for r in got) > 20 * len(short)


def test_nld_and_large_alphabet(data_dir, tmp_path):
    """nld.aspell (d = 3, strings up to 24 symbols: the 8-word kernels) and an alphabet of 70 classes (count-vector scan tiles)."""
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "nld.aspell.lexicon"))
    g.build()
    words = synth.load_lexicon_words(os.path.join(data_dir, "nld.aspell.lexicon"))
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=3, max_matches=10)
    qs = synth.make_queries(words, 1500, max_len=24, seed=8)
    t0 = small_stats()
    got = call(g, qs, p)
    assert small_stats()[0] == t0[0] + 1 and got == via_batch_path(g, qs, p)
    # 70 alphabet classes: beyond the 32 symbols of the bit-plane scan
    syms = [chr(c) for c in range(0x430, 0x430 + 32)] + [chr(c) for c in range(ord("a"), ord("z") + 1)] + list("0123456789") + ["-", "'"]
    alpha = tmp_path / "big.alphabet.tsv"
    alpha.write_text("".join(f"{s}\n" for s in syms[:70]), encoding="utf-8")
    rng = random.Random(4)
    lex = tmp_path / "big.lexicon"
    vocab = sorted({"".join(rng.choice(syms[:70]) for _ in range(rng.randrange(3, 12))) for _ in range(30000)})
    lex.write_text("".join(f"{w}\n" for w in vocab), encoding="utf-8")
    g2 = A.VariantModel(str(alpha), A.Weights(), device=0)
    g2.read_lexicon(str(lex))
    g2.build()
    qs2 = synth.make_queries(vocab, 900, max_len=12, seed=5)
    p2 = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    t0 = small_stats()
    got2 = call(g2, qs2, p2)
    assert sum(small_stats()) == sum(t0) + 1 and got2 == via_batch_path(g2, qs2, p2)
    assert sum(len(r) for r in got2) > 500


def test_concurrent_small_calls(eng):
    """Eight host threads, each issuing calls of 1 .. 1000 inputs on the one model (a context per call in flight)."""
    g, o, words = eng
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    op = O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0)
    sets = [synth.make_queries(words, n, max_len=16, seed=900 + i) for i, n in enumerate((1, 1000, 37, 512, 3, 1000, 250, 64))]
    want = [via_batch_path(g, qs, p) for qs in sets]
    errors = []

    def work(i):
        try:
            for _ in range(30):
                if call(g, sets[i], p) != want[i]:
                    errors.append(i)
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(sets))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert want[0][0] == o.find_variants(sets[0][0], op)
