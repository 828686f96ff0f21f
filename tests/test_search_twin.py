"""Search mode (find_all_matches) of the oracle twin, pinned by the reference's own tests
(/root/reference/tests/main.rs 06xx boundaries / n-grams, 07xx find_all_matches incl. bigram LM; values transcribed)
and tutorial.ipynb cells 22/24."""
import os

import pytest

from oracle import twin as T

A = T.TEST_ALPHABET


def test0601_find_boundaries():  # :1023-1040
    text = 'Hallo allemaal, ik zeg: "Welkom in Aix-les-bains!".'
    b = T.find_boundaries(text)
    assert len(b) == 9 and (b[0].begin, b[0].end, b[0].text) == (5, 6, " ")
    assert [x.text for x in b[1:]] == [", ", " ", ': "', " ", " ", "-", "-", '!".']


@pytest.mark.parametrize("text,order,exp", [
    ("dit is een mooie test", 1, ["dit", "is", "een", "mooie", "test"]),          # :1043-1053
    ("dit is een mooie test.", 1, ["dit", "is", "een", "mooie", "test"]),         # :1056-1066
    ("hello, world!", 1, ["hello", "world"]),                                       # :1069-1076
    ("dit is een mooie test.", 2, ["dit is", "is een", "een mooie", "mooie test"]),  # :1079-1092
    ("hello,world!", 2, ["hello,world"]),                                           # :1095-1101
    ("hello, world!", 2, ["hello, world"]),                                         # :1104-1110
    ("hello!", 2, []),                                                               # :1113-1118
])
def test060x_find_ngrams(text, order, exp):
    assert [m.text for m in T.find_match_ngrams(text, T.find_boundaries(text), order, 0, None)] == exp


def test0605_boundary_count():
    assert len(T.find_boundaries("dit is een mooie test.")) == 5


def _lm_model(extra_words=(), extra_lm=()):
    m = T.SearchModel(A)
    for w in ("I", "think", "sink", "you", "are", "right") + tuple(extra_words) + ("are right",):
        m.add_to_vocabulary(w, 2)
    for t, f in (("<bos> I", 2), ("I think", 2), ("I sink", 1), ("you are", 2), ("right <eos>", 2)) + tuple(extra_lm):
        m.add_lm(t, f)
    m.build()
    return m


def test0701_unigram_only():  # :1121-1141
    m = T.SearchModel(A)
    for w in ("I", "think", "sink", "you", "are", "right"):
        m.add_to_vocabulary(w)
    m.build()
    p = T.test_searchparams_search()
    p.max_ngram = 1
    r = m.find_all_matches("I tink you are rihgt", p)
    assert [x.text for x in r] == ["I", "tink", "you", "are", "rihgt"]
    assert m.match_to_str(r[1]) == "think" and m.match_to_str(r[4]) == "right"


def test0702_find_all_matches_lm():  # :1144-1206
    m = _lm_model()
    r = m.find_all_matches("I tink you are rihgt", T.test_searchparams_search())
    assert [(x.text, m.match_to_str(x)) for x in r] == [("I", "I"), ("tink", "think"), ("you", "you"), ("are rihgt", "are right")]
    assert (r[1].begin, r[1].end) == (2, 6)


def test0703_linebreak():  # :1209-1266
    m = _lm_model()
    r = m.find_all_matches("I tink you are\nrihgt", T.test_searchparams_search())
    assert [(x.text, m.match_to_str(x)) for x in r] == [("I", "I"), ("tink", "think"), ("you", "you"), ("are\nrihgt", "are right")]


def test0704_two_batches():  # :1269-1362
    m = _lm_model(("am", "sure"), (("I am", 2), ("sure <eos>", 2)))
    r = m.find_all_matches("I tink you are rihgt\n\nI am sur", T.test_searchparams_search())
    assert [(x.text, m.match_to_str(x)) for x in r] == [("I", "I"), ("tink", "think"), ("you", "you"), ("are rihgt", "are right"),
                                                         ("I", "I"), ("am", "am"), ("sur", "sure")]


def test0705_lm_disabled():  # :1365-1424 (lm_weight = 0)
    m = _lm_model()
    p = T.test_searchparams_search()
    p.lm_weight = 0.0
    r = m.find_all_matches("I tink you are rihgt", p)
    assert [(x.text, m.match_to_str(x)) for x in r] == [("I", "I"), ("tink", "think"), ("you", "you"), ("are rihgt", "are right")]


def test0706_0707_offsets():  # :1427-1481
    m = T.SearchModel(A)
    for w in ("I", "think", "you", "are", "right"):
        m.add_to_vocabulary(w)
    m.build()
    p = T.test_searchparams_search()
    p.max_ngram = 1
    p.unicodeoffsets = True
    r = m.find_all_matches("I thиnk you are righт", p)
    assert [x.text for x in r] == ["I", "thиnk", "you", "are", "righт"]
    assert (r[1].begin, r[1].end) == (2, 7) and m.match_to_str(r[1]) == "think" and m.match_to_str(r[4]) == "right"
    p.unicodeoffsets = False
    r = m.find_all_matches("I thиnk you are rihgt", p)
    assert (r[1].begin, r[1].end) == (2, 8) and m.match_to_str(r[4]) == "right"


def test_tutorial_find_all_matches(data_dir, tutorial_outputs):
    """tutorial.ipynb cells 22 and 24: eng.aspell, SearchParameters(unicodeoffsets=True) (max_ngram 3, no LM)."""
    al = T.read_alphabet(os.path.join(data_dir, "simple.alphabet.tsv"))
    m = T.SearchModel(al)
    m.read_vocabulary(os.path.join(data_dir, "eng.aspell.lexicon"))
    m.build()
    p = T.SearchParams(unicodeoffsets=True)
    case = tutorial_outputs["find_all_matches"][0]
    r = m.find_all_matches(case["input"], p)
    assert [(x.text, x.begin, x.end) for x in r] == [(c["input"], c["begin"], c["end"]) for c in case["matches"]]
    for x, c in zip(r, case["matches"]):
        assert [[m.decoder[v.vocab_id].text, v.score(0.0), v.dist_score, v.freq_score] for v in x.variants] == c["variants"]
    case = tutorial_outputs["find_all_matches"][1]
    r = m.find_all_matches(case["input"], p)
    c = case["matches"][0]
    x = r[case["only_match_index"]]
    assert (x.text, x.begin, x.end) == (c["input"], c["begin"], c["end"])
    assert [[m.decoder[v.vocab_id].text, v.score(0.0), v.dist_score, v.freq_score] for v in x.variants] == c["variants"]


# -- context rules: tests/main.rs:1575-1800 (values transcribed) -------------------------------------------------
def _rules_model():
    m = T.SearchModel(A)
    for w in ("I", "think", "sink", "you", "are", "right"):  # "context rule will decide between think/sink"
        m.add_to_vocabulary(w, 2)
    m.build()
    return m


def _rules_params():
    p = T.test_searchparams_search()
    p.lm_weight = 0.0
    p.max_ngram = 1
    return p


def test0902_context_rules_bonus():  # :1575-1610
    m = _rules_model()
    m.add_contextrule("I; think", 1.1, ["testtag"], [])
    r = m.find_all_matches("I tink you are rihgt", _rules_params())
    assert [x.text for x in r] == ["I", "tink", "you", "are", "rihgt"]
    assert [m.match_to_str(x) for x in r] == ["I", "think", "you", "are", "right"]
    assert (r[0].tag, r[0].seqnr, r[1].tag, r[1].seqnr) == ([0], [0], [0], [1])


def test0903_context_rules_penalty():  # :1613-1640
    m = _rules_model()
    m.add_contextrule("I; think", 0.9, [], [])
    r = m.find_all_matches("I tink you are rihgt", _rules_params())
    assert [m.match_to_str(x) for x in r] == ["I", "sink", "you", "are", "right"]


def test0904_context_rules_tags():  # :1643-1685
    m = _rules_model()
    for w in ("think", "are", "right"):
        m.add_contextrule(w, 1.0, ["testtag"], [])
    r = m.find_all_matches("I tink you are rihgt", _rules_params())
    assert [m.match_to_str(x) for x in r] == ["I", "think", "you", "are", "right"]
    assert [x.tag for x in r] == [[], [0], [], [0], [0]]
    assert [x.seqnr for x in r] == [[], [0], [], [0], [0]]


def test0905_context_rules_multitag():  # :1688-1722
    m = _rules_model()
    m.add_contextrule("I; think", 1.1, ["testtag", "testtag2"], [])
    r = m.find_all_matches("I tink you are rihgt", _rules_params())
    assert [m.match_to_str(x) for x in r] == ["I", "think", "you", "are", "right"]
    assert (r[0].tag, r[0].seqnr, r[1].tag, r[1].seqnr) == ([0, 1], [0, 0], [0, 1], [1, 1])
    assert m.tags == ["testtag", "testtag2"]


def test_pattern_parse_and_match():  # src/search.rs:373-459
    m = _rules_model()
    m.lexicons = ["data/a.tsv", "b.tsv"]
    P = lambda s: T.parse_pattern(s, m.lexicons, m.encoder)
    think = m.encoder["think"]
    assert P(" ? ") == ("any",) and P("^") == ("nolex",) and P("think") == ("vocab", think)
    assert P("@a.tsv") == ("lex", 0) and P("@b.tsv") == ("lex", 1) and P("@data/a.tsv") == ("lex", 0)
    assert P("!think") == ("not", ("vocab", think))
    assert P("think|sink")[0] == "or" and P("!(think|sink)")[0] == "not"
    assert P("!think|sink") == ("or", [("not", ("vocab", think)), ("vocab", m.encoder["sink"])])
    with pytest.raises(ValueError):
        P("unknownword")
    with pytest.raises(ValueError):
        P("@c.tsv")
    assert T.pattern_matches(P("^"), (0, 0)) and T.pattern_matches(P("^"), (7, 0)) and not T.pattern_matches(P("^"), (7, 1))
    assert T.pattern_matches(P("@b.tsv"), (7, 2)) and not T.pattern_matches(P("@b.tsv"), (7, 1))
    assert T.pattern_matches(P("!(think|sink)"), (9999, 1)) and not T.pattern_matches(P("!(think|sink)"), (think, 1))


def test_tagoffsets():  # src/lib.rs:703-751, src/search.rs:496-515
    m = _rules_model()
    m.add_contextrule("I; think; you", 1.2, ["a", "b"], ["1:1", ":2"])
    assert m.context_rules[0].tagoffset == [(1, 1), (0, 2)]
    r = m.find_all_matches("I tink you are rihgt", _rules_params())
    assert [x.tag for x in r] == [[1], [0, 1], [], [], []]
    assert [x.seqnr for x in r] == [[0], [0, 1], [], [], []]
    with pytest.raises(ValueError):
        m.add_contextrule("I", 1.0, ["t"], ["x:1"])
    with pytest.raises(ValueError):
        m.add_contextrule("I", 1.0, [""], [])
