"""GPU parity for confusable weighting: the reference's tests 0502-0504
(/root/reference/tests/main.rs:929-1020, values transcribed) and product vs twin on a small lexicon, late (default)
and early (set_confusables_before_pruning) rescoring."""
import os
import random

import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import twin as T

TEST_ALPHABET_TSV = "\n".join(f"{c}\t{c.upper()}" for c in "abcdefghijklmnopqrstuvwxyz") + "\n.\t,\n"


def _gp(**kw):  # src/test.rs:48-68
    d = dict(max_anagram_distance=2, max_edit_distance=2, max_matches=10, score_threshold=0.0, cutoff_threshold=0.0)
    d.update(kw)
    return A.SearchParameters(**d)


def _small(script):
    g = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=0)
    for w in ("huis", "huls"):
        g.add_to_vocabulary(w)
    g.add_to_confusables(script, 1.1)
    g.build()
    return g


def test0502_0503_confusable_boosts_huis():
    g = _small("-[y]+[i]")
    for q in ("huys", "Huys"):
        r = g.find_variants(q, _gp())
        assert [x["text"] for x in r] == ["huis", "huls"]
        assert r[0]["dist_score"] > r[1]["dist_score"]


def test0504_confusable_nomatch():
    r = _small("-[y]+[p]").find_variants("Huys", _gp())
    assert len(r) == 2 and r[0]["dist_score"] == r[1]["dist_score"]


@pytest.mark.parametrize("early", [False, True])
def test_random_vs_twin(early, tmp_path):
    words = [w for w in synth.load_lexicon_words(os.path.join(synth.GOLDEN_DATA, "eng_aspell.lexicon.gz"))
             if w.isascii() and w.isalpha()][::53][:2000]
    rng = random.Random(3)
    conf = tmp_path / "confusables.tsv"
    conf.write_text("-[y]+[i]\t1.1\n-[a]+[e]\t1.05\n=[c|k]-[s]\t0.9\n+[e]$\t0.95\n^-[k]\t0.8\n-[e]=[r]\n+[s]\t0.97\n", encoding="utf-8")
    tw = T.VariantModel(T.TEST_ALPHABET)
    g = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=0)
    for w in words:
        f = rng.randrange(1, 30)
        tw.add_to_vocabulary(w, f)
        g.add_to_vocabulary(w, f)
    tw.read_confusablelist(str(conf))
    g.read_confusablelist(str(conf))
    if early:
        tw.set_confusables_before_pruning()
        g.set_confusables_before_pruning()
    tw.build()
    g.build()
    qs = synth.make_queries(words, 300, max_len=16, seed=21)
    for fw, mm, cut in ((0.0, 5, 2.0), (0.5, 3, 1.5), (0.0, 0, 0.0)):
        gp = A.SearchParameters(max_anagram_distance=2, max_edit_distance=2, max_matches=mm, score_threshold=0.3,
                                cutoff_threshold=cut, freq_weight=fw)
        tp = T.SearchParameters(("abs", 2), ("abs", 2), mm, 0.3, cut, False, fw)
        got = g.find_variants_ids(qs, gp)
        changed = 0
        for q, r in zip(qs, got):
            exp = tw.find_variants(q, tp)
            assert [v for v, _d, _f in r] == [x.vocab_id for x in exp], (q, fw, mm)
            for (v, d, f), x in zip(r, exp):
                assert abs(d - x.dist_score) < 1e-6 and abs(f - x.freq_score) < 1e-6
            changed += any(tw.compute_confusable_weight(q, x.vocab_id) != 1.0 for x in exp)
        assert changed > 10  # the patterns did fire
