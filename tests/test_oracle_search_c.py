"""The C restatement of search mode (oracle/anx_oracle_search.inc; test infrastructure, the CPU baseline of BASELINE configs[4]) pinned by
the reference's own tests (/root/reference/tests/main.rs 07xx find_all_matches incl. the bigram LM; values transcribed, as in
tests/test_search_twin.py) and by the pure-Python twin (oracle/twin.py) on random running text with n-grams up to 3 and a bigram LM."""
import os
import random

import pytest

from analiticcl_amd import synth
from oracle import cwrap as O
from oracle import twin as T

TEST_TSV = "\n".join(f"{c}\t{c.upper()}" for c in "abcdefghijklmnopqrstuvwxyz") + "\n.\t,\n"


def _model(words, lm=(), freq=2):
    m = O.OracleModel(alphabet_text=TEST_TSV)
    for w in words:
        m.add(w, freq)
    for t, f in lm:
        m.add_lm(t, f)
    m.build()
    return m


def _sp(max_ngram=2, lm_weight=1.0):  # src/test.rs:48-68
    return O.make_search_params(O.make_params(("abs", 2), ("abs", 2), 10, 0.0, 0.0), max_ngram=max_ngram, lm_weight=lm_weight)


def _sel(m, res):
    return [(t, m.text(var[sel][0]) if (sel is not None and var) else t) for t, _b, _e, _n, sel, var in res]


LM = (("<bos> I", 2), ("I think", 2), ("I sink", 1), ("you are", 2), ("right <eos>", 2))
WORDS = ("I", "think", "sink", "you", "are", "right")


def test0701_unigram_only():  # tests/main.rs:1121-1141
    m = _model(WORDS, freq=None)
    res, _ = m.find_all_matches("I tink you are rihgt", _sp(max_ngram=1))
    assert _sel(m, res) == [("I", "I"), ("tink", "think"), ("you", "you"), ("are", "are"), ("rihgt", "right")]


def test0702_0703_0705_lm():  # :1144-1266, :1365-1424
    m = _model(WORDS + ("are right",), LM)
    exp = [("I", "I"), ("tink", "think"), ("you", "you"), ("are rihgt", "are right")]
    res, _ = m.find_all_matches("I tink you are rihgt", _sp())
    assert _sel(m, res) == exp and (res[1][1], res[1][2]) == (2, 6)
    res, _ = m.find_all_matches("I tink you are\nrihgt", _sp())
    assert _sel(m, res) == exp[:3] + [("are\nrihgt", "are right")]
    res, _ = m.find_all_matches("I tink you are rihgt", _sp(lm_weight=0.0))
    assert _sel(m, res) == exp


def test0704_two_batches():  # :1269-1362
    m = _model(WORDS + ("am", "sure", "are right"), LM + (("I am", 2), ("sure <eos>", 2)))
    res, _ = m.find_all_matches("I tink you are rihgt\n\nI am sur", _sp())
    assert _sel(m, res) == [("I", "I"), ("tink", "think"), ("you", "you"), ("are rihgt", "are right"), ("I", "I"), ("am", "am"), ("sur", "sure")]


def test0707_byte_offsets():  # :1454-1481
    m = _model(("I", "think", "you", "are", "right"), freq=None)
    res, _ = m.find_all_matches("I thиnk you are rihgt", _sp(max_ngram=1))
    assert (res[1][1], res[1][2]) == (2, 8) and _sel(m, res)[4] == ("rihgt", "right")


@pytest.fixture(scope="module")
def worlds(data_dir):
    """twin + C oracle over eng.aspell, once without and once with a bigram LM (the twin's vocabulary build takes ~20 s each)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from search_common import TwinOverOracle
    lex = os.path.join(data_dir, "eng.aspell.lexicon")
    words = synth.load_lexicon_words(lex)
    common = [w for w in words if w.isalpha()][::23][:5000]
    out = {}
    for with_lm in (False, True):
        rng = random.Random(7)
        lm = [(f"{rng.choice(common)} {rng.choice(common)}", rng.randrange(1, 20)) for _ in range(20000)] + [(f"<bos> {w}", 5) for w in common[:500]] if with_lm else []
        tw = TwinOverOracle(T.read_alphabet(os.path.join(data_dir, "simple.alphabet.tsv")))
        tw.read_vocabulary(lex)
        for t, f in lm:
            tw.add_lm(t, f)
        tw.build()
        om = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
        om.read_lexicon(lex)
        for t, f in lm:
            om.add_lm(t, f)
        om.build()
        tw.attach(om)   # the twin's own find_variants takes seconds per query on a 119 k-entry lexicon: answered by the C oracle (same results: tests/test_oracle_c.py)
        out[with_lm] = (tw, om, common)
    return out


@pytest.mark.parametrize("max_ngram,with_lm", [(1, False), (2, True), (3, True), (3, False)])
def test_equals_the_twin_on_running_text(worlds, max_ngram, with_lm):
    """eng.aspell, sentences of perturbed words (the generator of BASELINE configs[4]'s workload), a bigram LM: segmentation, variants, the
    chosen sequence -- everything the twin returns -- equal, text by text."""
    tw, om, common = worlds[with_lm]
    texts = synth.make_running_text(common, 0.03, seed=11 + max_ngram)[:7] + ["", "one", "the cat and the dgo", "a-b c_d e'f", "x\n\ny  z.", "Ünïcödé wörds hëre"]
    tp = T.SearchParams(("abs", 3), ("abs", 2), 10, 0.25, 2.0, False, 0.0, max_ngram=max_ngram)
    sp = O.make_search_params(O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0), max_ngram=max_ngram)
    nsel = 0
    for text in texts:
        exp = tw.find_all_matches(text, tp)
        got, _pairs = om.find_all_matches(text, sp)
        assert [(g[0], g[1], g[2], g[3]) for g in got] == [(e.text, e.begin, e.end, e.n) for e in exp], text
        for g, e in zip(got, exp):
            ev = None if e.variants is None else [(v.vocab_id, v.dist_score, v.freq_score) for v in e.variants]
            assert g[5] == ev, (text, e.text)
            assert g[4] == e.selected, (text, e.text, g[4], e.selected)
            nsel += e.selected is not None
    assert nsel > 25
    rc, counts, tm, _tr, _tp = om.find_all_matches_batch(texts, sp, nthreads=4)
    assert rc == 0 and tm == sum(counts) > 50
