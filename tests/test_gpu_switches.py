"""Exactness of the filters that decide whether a pair reaches damerau_levenshtein (/root/reference/src/distance.rs:101-179) at all,
and of every A/B switch of the hot path (DESIGN.md "Switches").

1. The band-match bound by itself (anx_debug_band_bound runs kernels_swar.hpp's band_bound_rejects on the device, one lane per
   pair, in the three forms the kernels use): a rejected pair must have DL > d in the ORACLE, over every length pair <= 16,
   d <= 3, random strings over small alphabets (many equal symbols), edited copies and transposition-heavy strings.
2. Full size: BASELINE configs[1] (1 M queries) and configs[2]'s scan shape (nld, d = 3) run with the default path, with
   ANX_SCAN_FUSE=0 (the scan leaves the bound to k_filter_score) and with ANX_PREFILTER=0 (no bound anywhere: every
   length-compatible pair goes through the DL) must agree on the scored-pair count, the survivor count and the result checksum
   -- a false reject anywhere, also of a pair that would not have made the top n, changes n_survivors.
3. The remaining result-neutral switches (ANX_SCAN=sad, ANX_SCORE_FAST=0, ANX_FS_SPLIT=0, ANX_FS_B7=0, ANX_FS_PLANES=0) on a mid-size batch."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import _lib as L
from analiticcl_amd import synth
from oracle import cwrap as O

from .fullsize_common import checksum


def _rows(strings, pad):
    out = np.full((len(strings), 16), pad, dtype=np.uint8)
    lens = np.zeros(len(strings), dtype=np.uint8)
    for i, s in enumerate(strings):
        out[i, :len(s)] = s
        lens[i] = len(s)
    return out, lens


def _pairs(rng, n):
    """(q, c) symbol strings of 1..16 symbols: unrelated, edited copies (<= 4 edits), transposition-heavy, shifted copies."""
    qs, cs = [], []
    for i in range(n):
        sigma = int(rng.choice([2, 3, 5, 12, 30]))
        lq = int(rng.integers(1, 17))
        q = rng.integers(0, sigma, lq).tolist()
        kind = i % 4
        if kind == 0:
            c = rng.integers(0, sigma, int(rng.integers(max(1, lq - 3), min(16, lq + 3) + 1))).tolist()
        else:
            c = list(q)
            for _ in range(int(rng.integers(0, 5))):
                op = int(rng.integers(0, 4)) if kind != 2 else 3
                p = int(rng.integers(0, len(c) + 1))
                if op == 0 and len(c) > 1:
                    del c[min(p, len(c) - 1)]
                elif op == 1 and len(c) < 16:
                    c.insert(p, int(rng.integers(0, sigma)))
                elif op == 2 and c:
                    c[min(p, len(c) - 1)] = int(rng.integers(0, sigma))
                elif op == 3 and len(c) > 1:
                    j = min(p, len(c) - 2)
                    c[j], c[j + 1] = c[j + 1], c[j]
            if kind == 3 and len(c) < 16 and rng.random() < 0.5:
                c = [int(rng.integers(0, sigma))] + c     # everything shifted by one
        qs.append(q)
        cs.append(c[:16] if c else [0])
    return qs, cs


@pytest.mark.parametrize("form", [0, 1, 2])
def test_band_bound_never_rejects_a_pair_within_d(form):
    rng = np.random.default_rng(1234 + form)
    qs, cs = _pairs(rng, 120_000)
    # every length pair <= 16 at least once, identical strings, single symbols
    for lq in range(1, 17):
        for lc in range(1, 17):
            qs.append(rng.integers(0, 3, lq).tolist())
            cs.append(rng.integers(0, 3, lc).tolist())
    qrows, lq = _rows(qs, 0xFE)
    crows, lc = _rows(cs, 0xFF)
    n = len(qs)
    total_rej = 0
    for d in (0, 1, 2, 3):
        out = np.zeros(n, dtype=np.uint8)
        # the kernels only filter pairs whose lengths differ by <= d (the DL's own first test drops the others): same here
        rc = L.lib().anx_debug_band_bound(0, qrows.ctypes.data, crows.ctypes.data, lq.ctypes.data, lc.ctypes.data, n, d, form,
                                          out.ctypes.data)
        assert rc == 0, L.last_error()
        rej = np.nonzero(out)[0]
        total_rej += rej.size
        for i in rej:
            if abs(int(lq[i]) - int(lc[i])) > d:
                continue
            assert O.dl(qs[i], cs[i], d) is None, (form, d, qs[i], cs[i])
        # and the bound is not vacuous: it rejects a good share of the unrelated pairs (many of them are strings over 2 or 3 symbols)
        unrelated = np.arange(0, 120_000, 4)
        assert out[unrelated].mean() > 0.15
    assert total_rej > n


def _run(b):
    b.run()
    st = b.stats()
    return (st["n_pairs"], st["n_survivors"], checksum(*b.fetch_arrays())), st


@pytest.mark.parametrize("lex,maxlen,d,nq", [("eng", 16, 2, 1_000_000), ("nld", 24, 3, 400_000)])
def test_filters_off_equal_default_at_full_size(data_dir, lex, maxlen, d, nq):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, f"{lex}.aspell.lexicon"))
    g.build()
    qs = synth.make_queries(synth.load_lexicon_words(os.path.join(data_dir, f"{lex}.aspell.lexicon")), nq, max_len=maxlen, seed=synth.SEED)
    b = g.encode_batch(qs, A.SearchParameters(max_anagram_distance=3, max_edit_distance=d, max_matches=10))
    try:
        ref, st0 = _run(b)
        assert st0["n_prefiltered_in_scan"] > 0.5 * st0["n_pairs"]   # the default path does filter in the scan
        A.set_switch("ANX_SCAN_FUSE", "0")
        fuse0, st1 = _run(b)
        assert st1["n_prefiltered_in_scan"] == 0 and st1["n_pair_slots"] > 1.5 * st0["n_pair_slots"]
        A.set_switch("ANX_SCAN_FUSE", None)
        A.set_switch("ANX_PREFILTER", "0")
        pre0, st2 = _run(b)
        assert st2["n_selected"] > 1.3 * st0["n_selected"]           # every length-compatible pair went through the DL (d = 3: the bound is weaker)
        assert fuse0 == ref and pre0 == ref
    finally:
        A.set_switch("ANX_SCAN_FUSE", None)
        A.set_switch("ANX_PREFILTER", None)
        b.free()


@pytest.mark.parametrize("switch,value", [("ANX_SCAN_ADJ", "0"), ("ANX_SCAN", "sad"), ("ANX_SCORE_FAST", "0"), ("ANX_FS_SPLIT", "0"), ("ANX_FS_B7", "0"), ("ANX_FS_PLANES", "0")])
def test_result_neutral_switches(data_dir, switch, value):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    g.build()
    qs = synth.make_queries(synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon")), 60_000, max_len=24, seed=11)
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    b = g.encode_batch(qs, p)
    ref, _ = _run(b)
    b.free()
    try:
        A.set_switch(switch, value)
        b = g.encode_batch(qs, p)   # ANX_SCAN is read when the tiles are built
        got, _ = _run(b)
        b.free()
        assert got == ref
    finally:
        A.set_switch(switch, None)


def _model(data_dir, lex):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, f"{lex}.aspell.lexicon"))
    g.build()
    return g


@pytest.mark.parametrize("lex,maxlen,k,d,nq", [("eng", 16, 3, 2, 1_000_000), ("nld", 24, 3, 3, 400_000), ("eng", 16, 2, 2, 200_000), ("eng", 12, 1, 1, 200_000)])
def test_adjacency_lists_equal_the_probe_walk_at_full_size(data_dir, lex, maxlen, k, d, nq):
    """k_scan_adj (tiles stream the prebuilt adjacency list of their signature, analiticcl_amd/csrc/adjacency.h) against
    ANX_SCAN_ADJ=0 (every tile enumerates its signature ball itself, as until round 4): scored pairs
    (= find_nearest_anahashes' candidates x instances, /root/reference/src/lib.rs:1143-1402), survivors and result checksum."""
    g = _model(data_dir, lex)
    qs = synth.make_queries(synth.load_lexicon_words(os.path.join(data_dir, f"{lex}.aspell.lexicon")), nq, max_len=maxlen, seed=synth.SEED + k)
    p = A.SearchParameters(max_anagram_distance=k, max_edit_distance=d, max_matches=10)
    try:
        b = g.encode_batch(qs, p)
        got, st = _run(b)
        b.free()
        assert st["n_adj_tiles"] > 0.9 * st["n_scan_blocks"]     # nearly every tile has a list (closure 2 of the lexicon's signatures)
        A.set_switch("ANX_SCAN_ADJ", "0")
        b = g.encode_batch(qs, p)   # read when the tiles are built
        ref, st0 = _run(b)
        b.free()
        assert st0["n_adj_tiles"] == 0
        assert got == ref
    finally:
        A.set_switch("ANX_SCAN_ADJ", None)


def test_adjacency_mixed_with_probe_tiles_stop_at_exact_and_pair_counts(data_dir):
    """Lists for the lexicon's own signatures only (ANX_ADJ_CLOSURE=0: a third of the tiles keep the probe walk, both scan kernels run in
    one batch), the general kernel instance (StopAtExactMatch) and the per-query pair counts of the production run -- all against the
    oracle's find_variants on a sample and against the run without lists."""
    qs = synth.make_queries(synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon")), 120_000, max_len=20, seed=77)
    try:
        A.set_switch("ANX_ADJ_CLOSURE", "0")
        g = _model(data_dir, "eng")   # read when the model goes to the device
        o = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
        o.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
        o.build()
        for stop in (False, True):
            p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, stop_criterion=stop)
            op = O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0, stop_at_exact_match=stop)
            b = g.encode_batch(qs, p)
            got, st = _run(b)
            assert 0.3 * st["n_scan_blocks"] < st["n_adj_tiles"] < 0.95 * st["n_scan_blocks"]
            counts = b.pair_counts()
            rows = b.fetch()
            b.free()
            for i in range(0, len(qs), 997):
                exp = o.find_variants(qs[i], op)
                assert [(v, dd, f) for v, dd, f in rows[i]] == exp, (stop, qs[i])
            A.set_switch("ANX_SCAN_ADJ", "0")
            b = g.encode_batch(qs, p)
            ref, _ = _run(b)
            counts0 = b.pair_counts()
            b.free()
            A.set_switch("ANX_SCAN_ADJ", None)
            assert got == ref and np.array_equal(counts, counts0)
    finally:
        A.set_switch("ANX_SCAN_ADJ", None)
        A.set_switch("ANX_ADJ_CLOSURE", None)
