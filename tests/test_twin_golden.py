"""Pins oracle/twin.py against the reference's own known answers.

Sources (values transcribed, not code):
  * /root/reference/tests/main.rs  tests 01xx (hash), 02xx (iterators), 03xx (distances), 04xx (model)
  * /root/reference/tutorial.ipynb recorded outputs -> tests/golden/tutorial_outputs.json
"""
import os

import pytest

from oracle import twin as T

A = T.TEST_ALPHABET
ASZ = 27  # get_test_alphabet() returns alphabet.len() (no UNK) as alphabet_size, src/test.rs:44-45


def h(s):
    return T.anahash(s, A)


# ---- 00xx / 01xx ------------------------------------------------------------------------------
def test0001_alphabet():
    assert len(A) == 27


def test0002_primes():
    assert len(T.PRIMES) == 168
    for p in T.PRIMES:
        assert all(p % i for i in range(2, p))


def test0103_hash_basic():  # tests/main.rs:38-55
    assert h("a") == 2 and h("b") == 3 and h("c") == 5
    assert h("ab") == 6 == h("ba")
    assert h("abc") == 30
    assert h("abcabcabc") == 2 * 3 * 5 * 2 * 3 * 5 * 2 * 3 * 5
    assert T.anahash("", A) == 1  # tests/main.rs:30-35


def test0103_hash_alphabet_equivalence():  # :58-68
    assert h("abc") == h("ABC") == h("bAc")
    assert h("a.b") == h("a,b")


def test0104_hash_big():  # :71-80
    assert h("xyz" * 24) > 1


def test0105_hash_anagram():  # :83-91
    assert h("stressed") == h("desserts")
    assert h("dormitory") == h("dirtyroom")
    assert h("presents") == h("serpents")


def test0106_insertion():  # :94-103
    assert T.av_insert(h("ab"), h("c")) == h("abc") == T.av_insert(h("c"), h("ab"))


def test0107_containment():  # :106-121
    ab, c, abc = h("ab"), h("c"), h("abc")
    assert T.av_contains(abc, c) and T.av_contains(abc, ab) and T.av_contains(abc, abc)
    assert not T.av_contains(c, abc) and not T.av_contains(ab, c) and not T.av_contains(ab, abc)


def test0108_deletion():  # :124-140
    assert T.av_delete(h("abc"), h("c")) == h("ab")
    assert T.av_delete(h("abc"), h("b")) == h("ac")
    assert T.av_delete(h("c"), h("abc")) is None
    assert T.av_delete(h("abc"), h("x")) is None


def test0108_upper_bound():  # :143-153
    assert T.alphabet_upper_bound(h("abc"), ASZ) == (2, 3)
    assert T.alphabet_upper_bound(h("ab"), ASZ) == (1, 2)
    assert T.alphabet_upper_bound(h("x"), ASZ) == (23, 1)


# ---- 02xx iterators --------------------------------------------------------------------------
def test0201_iterator_parents():  # :156-176
    res = list(T.av_iter_parents(h("house"), ASZ))
    assert [T.av_character(c) for _, c in res] == [h(x) for x in "usohe"]
    assert [v for v, _ in res] == [h(x) for x in ("hose", "houe", "huse", "ouse", "hous")]


def test0202_iterator_parents_dup():  # :179-198
    res = list(T.av_iter_parents(h("pass"), ASZ))
    assert [T.av_character(c) for _, c in res] == [h(x) for x in "spa"]
    assert [v for v, _ in res] == [h(x) for x in ("pas", "ass", "pss")]


def test0203_singlebeam():  # :201-224
    res = list(T.av_iter(h("house"), ASZ))
    assert [T.av_character(n[1]) for n, _ in res] == [h(x) for x in "usohe"]
    assert [n[0] for n, _ in res] == [h("hose"), h("hoe"), h("he"), h("e"), 1]
    assert [d for _, d in res] == [1, 2, 3, 4, 5]


def _vals(it, n=None):
    out = [(node[0], depth) for node, depth in it]
    return out if n is None else out[:n]


def test0203_recursive_dfs():  # :227-258
    got = [v for v, _ in _vals(T.av_iter_recursive(h("abcd"), ASZ), 19)]
    exp = ["abc", "ab", "a", "", "b", "", "ac", "a", "", "c", "", "bc", "b", "", "c", "", "abd", "ab", "a"]
    assert got == [h(x) for x in exp]


def test0203_recursive_no_empty_leaves():  # :261-292
    got = [v for v, _ in _vals(T.av_iter_recursive(h("abcd"), ASZ, allow_empty_leaves=False), 13)]
    exp = ["abc", "ab", "a", "b", "ac", "a", "c", "bc", "b", "c", "abd", "ab", "a"]
    assert got == [h(x) for x in exp]


def test0203_recursive_no_duplicates():  # :295-322
    got = [v for v, _ in _vals(T.av_iter_recursive(h("abcd"), ASZ, allow_empty_leaves=False,
                                                  allow_duplicates=False), 8)]
    assert got == [h(x) for x in ["abc", "ab", "a", "b", "ac", "c", "bc", "abd"]]


def test0203_recursive_bfs():  # :325-392
    got = _vals(T.av_iter_recursive(h("abcd"), ASZ, breadthfirst=True), 20)
    exp = [("abc", 1), ("abd", 1), ("acd", 1), ("bcd", 1),
           ("ab", 2), ("ac", 2), ("bc", 2), ("ab", 2), ("ad", 2), ("bd", 2),
           ("ac", 2), ("ad", 2), ("cd", 2), ("bc", 2), ("bd", 2), ("cd", 2),
           ("a", 3), ("b", 3), ("a", 3), ("c", 3)]
    assert got == [(h(x), d) for x, d in exp]


BFS_UNIQUE = [("abc", 1), ("abd", 1), ("acd", 1), ("bcd", 1), ("ab", 2), ("ac", 2), ("bc", 2),
              ("ad", 2), ("bd", 2), ("cd", 2), ("a", 3), ("b", 3), ("c", 3), ("d", 3)]


def test0203_bfs_no_duplicates():  # :395-449
    got = _vals(T.av_iter_recursive(h("abcd"), ASZ, breadthfirst=True, allow_duplicates=False,
                                    allow_empty_leaves=False))
    assert got == [(h(x), d) for x, d in BFS_UNIQUE]


def test0203_bfs_max_dist():  # :452-507
    got = _vals(T.av_iter_recursive(h("abcd"), ASZ, breadthfirst=True, allow_duplicates=False,
                                    allow_empty_leaves=False, max_distance=3))
    assert got == [(h(x), d) for x, d in BFS_UNIQUE]


def test0203_bfs_max_dist2():  # :510-556
    got = _vals(T.av_iter_recursive(h("abcd"), ASZ, breadthfirst=True, allow_duplicates=False,
                                    allow_empty_leaves=False, max_distance=2))
    assert got == [(h(x), d) for x, d in BFS_UNIQUE[:10]]


# ---- 03xx distances --------------------------------------------------------------------------
def n(s):
    return T.normalize_to_alphabet(s, A)


def test0301_normalize():  # :559-563
    assert n("a") == [0] and n("b") == [1]


def test0302_levenshtein():  # :566-629
    for a, b, e in [("a", "a", 0), ("a", "b", 1), ("ab", "ac", 1), ("a", "ab", 1), ("ab", "a", 1),
                    ("ab", "ba", 2), ("abc", "xyz", 3)]:
        assert T.levenshtein(n(a), n(b), 99) == e


def test0303_damerau_levenshtein():  # :632-708
    for a, b, e in [("a", "a", 0), ("a", "b", 1), ("ab", "ac", 1), ("a", "ab", 1), ("ab", "a", 1),
                    ("ab", "ba", 1), ("abc", "xyz", 3), ("hipotesis", "hypothesis", 2)]:
        assert T.damerau_levenshtein(n(a), n(b), 99) == e
    # unrestricted DL, not OSA: ca -> abc = 2
    assert T.damerau_levenshtein(n("ca"), n("abc"), 99) == 2
    assert T.damerau_levenshtein(n("abc"), n("xyz"), 2) is None


def test0304_lcs_prefix_suffix():  # :711-807
    assert T.longest_common_substring_length(n("test"), n("testable")) == 4
    assert T.longest_common_substring_length(n("fasttest"), n("testable")) == 4
    assert T.longest_common_substring_length(n("abcdefhij"), n("def")) == 3
    assert T.longest_common_substring_length(n("def"), n("abcdefhij")) == 3
    assert T.common_prefix_length(n("test"), n("testable")) == 4
    assert T.common_prefix_length(n("testable"), n("test")) == 4
    assert T.common_prefix_length(n("fasttest"), n("testable")) == 0
    assert T.common_prefix_length(n("fasttest"), n("test")) == 0
    assert T.common_suffix_length(n("test"), n("testable")) == 0
    assert T.common_suffix_length(n("testable"), n("test")) == 0
    assert T.common_suffix_length(n("fasttest"), n("testable")) == 0
    assert T.common_suffix_length(n("fasttest"), n("test")) == 4


# ---- 04xx model -------------------------------------------------------------------------------
LEX = ["rites", "tiers", "tires", "tries", "tyres", "rides", "brides", "dire"]


def _model(words):
    m = T.VariantModel(A)
    for w in words:
        m.add_to_vocabulary(w, None)
    m.build()
    return m


def test0401_0402_model():  # :816-855
    m = _model(LEX)
    assert all(m.has(w) for w in LEX) and not m.has("unknown")
    assert [i.text for i in m.get_anagram_instances("rites")] == ["rites", "tiers", "tires", "tries"]
    m.find_variants("rite", T.test_searchparams())  # :858-869 (must not crash)


def test0404_score_test():  # :872-911
    m = _model(["huis", "huls"])
    r = m.find_variants("huys", T.test_searchparams())
    assert [m.decoder[x.vocab_id].text for x in r] == ["huis", "huls"]
    assert r[0].dist_score == r[1].dist_score and r[0].freq_score == r[1].freq_score


# ---- tutorial.ipynb recorded outputs (eng.aspell + simple.alphabet) -------------------------------
@pytest.fixture(scope="module")
def eng_model(data_dir):
    alphabet = T.read_alphabet(os.path.join(data_dir, "simple.alphabet.tsv"))
    m = T.VariantModel(alphabet)
    m.read_vocabulary(os.path.join(data_dir, "eng.aspell.lexicon"))
    m.build()
    return m


def test_tutorial_build_histogram(eng_model, tutorial_outputs):
    b = tutorial_outputs["build"]
    assert sum(len(v[0]) for v in eng_model.index.values()) == b["instances"]
    assert len(eng_model.index) == b["anagrams"]
    assert {int(k): len(v) for k, v in eng_model.sortedindex.items()} == {int(k): v for k, v in b["histogram"].items()}


def test_tutorial_find_variants(eng_model, tutorial_outputs):
    for case in tutorial_outputs["find_variants"]:
        res = eng_model.find_variants(case["input"], T.SearchParameters())
        got = [[eng_model.decoder[r.vocab_id].text, r.score(0.0), r.dist_score, r.freq_score] for r in res]
        assert got == case["results"], case["input"]


def test_tutorial_find_all_matches_unigrams(eng_model, tutorial_outputs):
    # the per-token variant lists printed by find_all_matches are find_variants(token) outputs
    case = tutorial_outputs["find_all_matches"][0]
    for m in case["matches"]:
        res = eng_model.find_variants(m["input"], T.SearchParameters())
        got = [[eng_model.decoder[r.vocab_id].text, r.score(0.0), r.dist_score, r.freq_score] for r in res]
        assert got == m["variants"], m["input"]
    # the bigram segment "sep arate" (tutorial cell 24): query string contains a space
    m = tutorial_outputs["find_all_matches"][1]["matches"][0]
    res = eng_model.find_variants(m["input"], T.SearchParameters())
    got = [[eng_model.decoder[r.vocab_id].text, r.score(0.0), r.dist_score, r.freq_score] for r in res]
    assert got == m["variants"]
