"""Pins the C oracle (oracle/anx_oracle.c) against
  (1) the reference's known-answer tests (/root/reference/tests/main.rs 01xx-04xx; values transcribed),
  (2) the reference's recorded outputs (tutorial.ipynb -> tests/golden/tutorial_outputs.json),
  (3) the independent Python twin on random inputs (differential).
"""
import os
import random

import pytest

from oracle import cwrap as O
from oracle import twin as T

TEST_ALPHABET_TSV = "\n".join(f"{c}\t{c.upper()}" for c in "abcdefghijklmnopqrstuvwxyz") + "\n.\t,\n"
ASZ = 27


@pytest.fixture(scope="module")
def tm():
    return O.OracleModel(alphabet_text=TEST_ALPHABET_TSV)


def test_alphabet_and_hash(tm):  # tests/main.rs:13-16, 38-91
    assert O.lib().orc_alphabet_len(tm.h) == 27
    h = tm.anahash
    assert (h("a"), h("b"), h("c"), h("ab"), h("ba"), h("abc")) == (2, 3, 5, 6, 6, 30)
    assert h("abcabcabc") == 30 ** 3 and h("") == 1
    assert h("abc") == h("ABC") == h("bAc") and h("a.b") == h("a,b")
    assert h("xyz" * 24) == (89 * 97 * 101) ** 24  # test0104: big value, exact
    assert h("stressed") == h("desserts") and h("dormitory") == h("dirtyroom") and h("presents") == h("serpents")


def test_contains_delete_upper_bound(tm):  # :106-153
    assert tm.contains("abc", "c") and tm.contains("abc", "ab") and tm.contains("abc", "abc")
    assert not tm.contains("c", "abc") and not tm.contains("ab", "c") and not tm.contains("ab", "abc")
    assert tm.upper_bound("abc", ASZ) == (2, 3)
    assert tm.upper_bound("ab", ASZ) == (1, 2)
    assert tm.upper_bound("x", ASZ) == (23, 1)


def test_iterators(tm):  # :156-556
    h = tm.anahash
    res = tm.iter_parents("house", ASZ)
    assert [T.PRIMES[c] for _, _, c in res] == [h(x) for x in "usohe"]
    assert [v for v, _, _ in res] == [h(x) for x in ("hose", "houe", "huse", "ouse", "hous")]
    res = tm.iter_parents("pass", ASZ)
    assert [v for v, _, _ in res] == [h(x) for x in ("pas", "ass", "pss")]
    res = tm.iter_recursive("house", ASZ, singlebeam=True)
    assert [v for v, _, _ in res] == [h("hose"), h("hoe"), h("he"), h("e"), 1]
    assert [d for _, d, _ in res] == [1, 2, 3, 4, 5]
    got = [v for v, _, _ in tm.iter_recursive("abcd", ASZ, max_items=19)]
    exp = ["abc", "ab", "a", "", "b", "", "ac", "a", "", "c", "", "bc", "b", "", "c", "", "abd", "ab", "a"]
    assert got == [h(x) for x in exp]
    got = [v for v, _, _ in tm.iter_recursive("abcd", ASZ, empty_leaves=False, max_items=13)]
    assert got == [h(x) for x in ["abc", "ab", "a", "b", "ac", "a", "c", "bc", "b", "c", "abd", "ab", "a"]]
    got = [v for v, _, _ in tm.iter_recursive("abcd", ASZ, empty_leaves=False, unique=True, max_items=8)]
    assert got == [h(x) for x in ["abc", "ab", "a", "b", "ac", "c", "bc", "abd"]]
    bfs = [("abc", 1), ("abd", 1), ("acd", 1), ("bcd", 1), ("ab", 2), ("ac", 2), ("bc", 2), ("ab", 2), ("ad", 2),
           ("bd", 2), ("ac", 2), ("ad", 2), ("cd", 2), ("bc", 2), ("bd", 2), ("cd", 2), ("a", 3), ("b", 3),
           ("a", 3), ("c", 3)]
    got = [(v, d) for v, d, _ in tm.iter_recursive("abcd", ASZ, breadthfirst=True, max_items=20)]
    assert got == [(h(x), d) for x, d in bfs]
    uniq = [("abc", 1), ("abd", 1), ("acd", 1), ("bcd", 1), ("ab", 2), ("ac", 2), ("bc", 2), ("ad", 2), ("bd", 2),
            ("cd", 2), ("a", 3), ("b", 3), ("c", 3), ("d", 3)]
    for maxd, exp in ((-1, uniq), (3, uniq), (2, uniq[:10])):
        got = [(v, d) for v, d, _ in tm.iter_recursive("abcd", ASZ, breadthfirst=True, unique=True,
                                                         empty_leaves=False, maxdepth=maxd)]
        assert got == [(h(x), d) for x, d in exp]


def test_distances(tm):  # :559-807
    n = tm.normalize
    assert n("a") == [0] and n("b") == [1]
    for a, b, e in [("a", "a", 0), ("a", "b", 1), ("ab", "ac", 1), ("a", "ab", 1), ("ab", "a", 1), ("ab", "ba", 2),
                    ("abc", "xyz", 3)]:
        assert O.lev(n(a), n(b), 99) == e
    for a, b, e in [("a", "a", 0), ("a", "b", 1), ("ab", "ac", 1), ("a", "ab", 1), ("ab", "a", 1), ("ab", "ba", 1),
                    ("abc", "xyz", 3), ("hipotesis", "hypothesis", 2), ("ca", "abc", 2)]:
        assert O.dl(n(a), n(b), 99) == e
    assert O.dl(n("abc"), n("xyz"), 2) is None
    assert O.lcs(n("test"), n("testable")) == 4 and O.lcs(n("fasttest"), n("testable")) == 4
    assert O.lcs(n("abcdefhij"), n("def")) == 3 and O.lcs(n("def"), n("abcdefhij")) == 3
    assert O.prefix(n("test"), n("testable")) == 4 and O.prefix(n("testable"), n("test")) == 4
    assert O.prefix(n("fasttest"), n("testable")) == 0 and O.prefix(n("fasttest"), n("test")) == 0
    assert O.suffix(n("test"), n("testable")) == 0 and O.suffix(n("testable"), n("test")) == 0
    assert O.suffix(n("fasttest"), n("testable")) == 0 and O.suffix(n("fasttest"), n("test")) == 4


def test_model_04xx():  # :816-911
    m = O.OracleModel(alphabet_text=TEST_ALPHABET_TSV)
    lex = ["rites", "tiers", "tires", "tries", "tyres", "rides", "brides", "dire"]
    for w in lex:
        m.add(w)
    m.build()
    assert all(m.has(w) for w in lex) and not m.has("unknown")
    assert m.anagram_instances("rites") == ["rites", "tiers", "tires", "tries"]
    p = O.make_params(("abs", 2), ("abs", 2), 10, 0.0, 0.0)
    m.find_variants("rite", p)
    m2 = O.OracleModel(alphabet_text=TEST_ALPHABET_TSV)
    for w in ("huis", "huls"):
        m2.add(w)
    m2.build()
    r = m2.find_variants("huys", p)
    assert [m2.text(v) for v, _, _ in r] == ["huis", "huls"]
    assert r[0][1] == r[1][1] and r[0][2] == r[1][2]


def test_distance_functions_vs_twin_random():
    rng = random.Random(7)
    for _ in range(20000):
        la, lb = rng.randrange(0, 12), rng.randrange(0, 12)
        a = [rng.randrange(4) for _ in range(la)]
        b = [rng.randrange(4) for _ in range(lb)]
        d = rng.randrange(0, 5)
        assert O.dl(a, b, d) == T.damerau_levenshtein(a, b, d)
        assert O.lev(a, b, d) == T.levenshtein(a, b, d)
        assert O.lcs(a, b) == T.longest_common_substring_length(a, b)
        assert O.prefix(a, b) == T.common_prefix_length(a, b)
        assert O.suffix(a, b) == T.common_suffix_length(a, b)


def test_clamp_threshold_vs_twin():
    for L in range(1, 80):
        for th in (("abs", 0), ("abs", 2), ("abs", 3), ("abs", 200), ("ratio", 0.3), ("ratio", 1.0), ("ratio", 0.0),
                   ("ratiolimit", 0.25, 3), ("ratiolimit", 0.9, 40)):
            assert O.lib().orc_clamp_threshold(O.threshold(th), L, 12) == T.clamp_threshold(th, L, 12)


@pytest.fixture(scope="module")
def eng(data_dir):
    m = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
    m.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    m.build()
    return m


def test_tutorial_build(eng, tutorial_outputs):
    b = tutorial_outputs["build"]
    assert eng.n_instances() == b["instances"] and eng.n_classes() == b["anagrams"]
    hist = {c: eng.bucket_size(c) for c in range(256) if eng.bucket_size(c)}
    assert hist == {int(k): v for k, v in b["histogram"].items()}


def test_tutorial_find_variants(eng, tutorial_outputs):
    p = O.make_params()  # SearchParameters::default(), src/types.rs:170-192
    cases = [(c["input"], c["results"]) for c in tutorial_outputs["find_variants"]]
    cases += [(m["input"], m["variants"]) for m in tutorial_outputs["find_all_matches"][0]["matches"]]
    cases += [(m["input"], m["variants"]) for m in tutorial_outputs["find_all_matches"][1]["matches"]]
    for text, exp in cases:
        got = [[eng.text(v), ds, ds, fs] for v, ds, fs in eng.find_variants(text, p)]
        assert got == exp, text


def test_eng_vs_twin_vectors(eng, twin_vectors):
    """The committed twin vectors (all parameter sets, incl. ratio thresholds, exact-stop, unlimited)."""
    ps = {k: O.make_params(tuple(v["max_anagram_distance"]), tuple(v["max_edit_distance"]), v["max_matches"],
                           v["score_threshold"], v["cutoff_threshold"], v["stop_at_exact_match"], v["freq_weight"])
          for k, v in twin_vectors["paramsets"].items()}
    for case in twin_vectors["cases"]:
        res, pairs, npairs, ncls = eng.find_variants(case["input"], ps[case["params"]], want_pairs=True)
        assert ncls == case["n_classes"] and npairs == case["n_pairs"], case
        got = [[eng.text(v), v, ds, fs] for v, ds, fs in res]
        assert got == case["results"], (case["params"], case["input"])
