"""The device-side query encoder (csrc/encode.hip: normalisation, count vectors, planes, signatures, clamps, sort, tiles, exact
class -- /root/reference/src/anahash.rs:16-80, src/lib.rs:982-1012, :1164-1173) against the threaded host encoder it replaces
(ANX_ENCODE=host, the A/B reference) and, through the results, against the oracle: identical rows, identical pair and class-test
counts (the latter only agree when queries are ordered and tiled identically), on inputs chosen for the encoder's corner
cases: multi-byte UTF-8, multi-character alphabet members, unknown characters, empty / over-long / invalid strings, ratio
thresholds, StopAtExactMatch, an alphabet of more than 32 symbols (count-vector scan), a 1 M batch."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import cwrap as O

SPECIAL = ["", "a", "''", "``quoted''", "it's", "naïve café", "Ünïcödé", "ßtraße", "ıstanbul", "ÉCOLE", "x" * 255, "y" * 256, "z" * 300,
           "tab\tsep", "dots...!!", "12345", "F", "K", "FK", "aaaaa", "aaaaaa", "eeeeeeeeee", "Mississippi", "\x1b", "a\x1bb",
           "日本語", "école", "\U0001F600 smile", "ǅ", "ᾈ", "seperate", "Seperate", "SEPERATE"]
RAW = [b"\xff\xfe", b"abc\xc3", b"\xe2\x82", b"ok\x80ok", b"\xf0\x9f\x98"]  # invalid / truncated UTF-8


def run_both(model, queries, params):
    out = {}
    for mode in ("device", "host"):
        A.set_switch("ANX_ENCODE", "host" if mode == "host" else None)
        try:
            b = model.encode_batch(queries, params)
            b.run()
            out[mode] = (b.fetch_arrays(), b.stats(), b.pair_counts())
            b.free()
        finally:
            A.set_switch("ANX_ENCODE", None)
    (da, ds, dc), (ha, hs, hc) = out["device"], out["host"]
    for x, y in zip(da, ha):
        assert np.array_equal(x, y)
    for k in ("n_queries", "n_pairs", "n_class_tests", "n_scan_blocks", "n_results", "n_survivors", "n_selected"):
        assert ds[k] == hs[k], k
    assert list(ds["n_tests_kind"]) == list(hs["n_tests_kind"])
    assert np.array_equal(dc, hc)
    return da, ds


@pytest.fixture(scope="module")
def eng(data_dir):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    g.build()
    return g


@pytest.mark.parametrize("kw", [dict(max_anagram_distance=3, max_edit_distance=2, max_matches=10),
                                dict(max_anagram_distance=3, max_edit_distance=2, max_matches=10, stop_criterion=True),
                                dict(max_anagram_distance=0.3, max_edit_distance=(0.25, 3), max_matches=5),
                                dict(max_anagram_distance=4, max_edit_distance=4, max_matches=0, cutoff_threshold=0.0)])
def test_device_equals_host_encoder(eng, data_dir, kw):
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    qs = SPECIAL + RAW + synth.make_queries(words, 20000, max_len=28, seed=41) + words[::997]
    (off, vid, dist, freq), st = run_both(eng, qs, A.SearchParameters(**kw))
    assert off.size == len(qs) + 1 and st["n_queries"] == len(qs) - 3   # "", 256 x y and 300 x z are not encodable
    assert off[1] == off[0]                      # "": no results


def test_special_inputs_vs_oracle(eng, data_dir):
    o = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
    o.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    o.build()
    gp = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    op = O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0)
    got = eng.find_variants_ids(SPECIAL, gp)
    for q, r in zip(SPECIAL, got):
        if q == "" or len(q) > 255:
            assert r == []
            continue
        assert [tuple(x) for x in r] == o.find_variants(q, op), q


def test_wide_alphabet_count_vector_scan(tmp_path):
    """More than 32 symbols: no thermometer planes, every query goes through the count-vector (SAD) tiles."""
    letters = "abcdefghijklmnopqrstuvwxyzäöüßéèêàçñøå0123456789"
    alpha = "".join(f"{c}\t{c.upper()}\n" if c.upper() != c and len(c.upper()) == 1 else f"{c}\n" for c in letters)
    g = A.VariantModel("", alphabet_text=alpha, device=0)
    import random
    rng = random.Random(3)
    words = sorted({"".join(rng.choice(letters) for _ in range(rng.randrange(3, 14))) for _ in range(4000)})
    for w in words:
        g.add_to_vocabulary(w)
    g.build()
    qs = synth.make_queries(words, 3000, max_len=16, seed=5) + ["Ärger", "ÀÉ", "ß9", ""]
    run_both(g, qs, A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10))


def test_million_queries_device_equals_host(eng, data_dir):
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    qs = synth.make_queries(words, 1_000_000, max_len=16, seed=synth.SEED)
    run_both(eng, qs, A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10))


@pytest.mark.parametrize("kw", [dict(max_anagram_distance=3, max_edit_distance=2, max_matches=10),
                                dict(max_anagram_distance=5, max_edit_distance=3, max_matches=10),
                                dict(max_anagram_distance=7, max_edit_distance=4, max_matches=5),
                                dict(max_anagram_distance=3, max_edit_distance=2, max_matches=10, stop_criterion=True)])
def test_hash_probe_walk_equals_flat_walk(eng, data_dir, monkeypatch, kw):
    """Candidate signatures two ways: enumerating the L1 ball of signature offsets and probing the hash table of the lexicon's
    signatures (default when the ball is smaller than the window) against the flat walk over the +-k charcount window
    (ANX_SCAN_WALK=flat).  Same pairs per query, same results; k = 7 has no ball (too large) and walks either way."""
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    qs = SPECIAL + synth.make_queries(words, 30000, max_len=28, seed=43) + ["a" * 130 + "b" * 125, "e" * 200]
    p = A.SearchParameters(**kw)
    out = {}
    for mode in ("probe", "flat"):
        A.set_switch("ANX_SCAN_WALK", "flat" if mode == "flat" else None)
        try:
            b = eng.encode_batch(qs, p)
            b.run()
            out[mode] = (b.fetch_arrays(), b.stats(), b.pair_counts())
            b.free()
        finally:
            A.set_switch("ANX_SCAN_WALK", None)
    (pa, ps, pc), (fa, fs, fc) = out["probe"], out["flat"]
    for x, y in zip(pa, fa):
        assert np.array_equal(x, y)
    assert np.array_equal(pc, fc)
    for k in ("n_queries", "n_pairs", "n_results", "n_survivors"):   # (n_class_tests counts chunk padding, which depends on the staging order)
        assert ps[k] == fs[k], k


def test_random_bytes_device_equals_host_encoder(eng):
    """Fuzz: 40 000 random byte strings (ASCII letters, alphabet punctuation, multi-character members, valid and broken UTF-8,
    lengths 1..300, no NUL) through both encoders: identical rows, pair counts and tiles."""
    import random
    rng = random.Random(2024)
    pieces = [bytes([c]) for c in range(1, 256)] + [b"''", b"``", "é".encode(), "ß".encode(), "日".encode(), "\U0001F600".encode(),
                                                     b"e", b"a", b"s", b"t", b"n", b"i", b"r", b"o", b" ", b".", b"-", b"'"] * 6
    qs = []
    for _ in range(40000):
        n = rng.choice((1, 2, 3, 5, 8, 8, 10, 12, 12, 16, 20, 30, 60, 150, 300))
        qs.append(b"".join(rng.choice(pieces) for _ in range(rng.randrange(1, n + 1))))
    run_both(eng, qs, A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10))


def test_packed_buffer_offsets_found_on_the_device(eng):
    """anx_batch_encode_packed hands the NUL-separated buffer to the device, which finds the strings itself (k_nul_count /
    k_nul_emit): empty strings, runs of NUL bytes, 0x01 bytes behind a NUL (the borrow case of a careless SWAR zero test),
    strings that straddle the 4096-byte blocks and the 16-byte loads, a buffer holding more strings than announced (the rest is
    ignored), one holding fewer (an error) -- against the char** entry point, whose offsets come from strlen on the host."""
    import random
    rng = random.Random(7)
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    words = ["seperate", "", "", "acommodate", "\x01", "\x01\x01x", "", "a" * 4090, "b" * 17, "", "c" * 4096, "d" * 15, "e" * 16, "wich", ""]
    words += ["".join(rng.choice("etaoinshr\x01 ") for _ in range(rng.randrange(0, 40))) for _ in range(30000)]
    want = eng.find_variants_ids(words, p)  # anx_find_variants_batch: char**
    packed = b"".join(w.encode() + b"\0" for w in words)
    for n in (len(words), len(words) - 1, 17, 1, 0):
        b = eng.encode_packed(packed, n, p)
        b.run()
        assert b.fetch() == want[:n], n
        b.free()
    with pytest.raises(A.AnxError, match="fewer strings than announced"):
        eng.encode_packed(packed, len(words) + 1, p)
    with pytest.raises(A.AnxError, match="must end with a NUL"):
        eng.encode_packed(packed[:-1], 3, p)
    # the host encoder (A/B) takes the same buffer through the host-side offset scan
    A.set_switch("ANX_ENCODE", "host")
    try:
        b = eng.encode_packed(packed, len(words), p)
        b.run()
        assert b.fetch() == want
        b.free()
        with pytest.raises(A.AnxError, match="fewer strings than announced"):
            eng.encode_packed(packed, len(words) + 1, p)
    finally:
        A.set_switch("ANX_ENCODE", None)
