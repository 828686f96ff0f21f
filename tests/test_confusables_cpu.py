"""Confusables in the oracle twin: the reference's tests 0501-0504 (/root/reference/tests/main.rs:914-1020, values
transcribed) and the published diff-match-patch behaviour the edit scripts are restated from (oracle/sesdiff_twin.py).
Parity beyond the four reference tests is UNPINNED (sesdiff / dissimilar are not vendored in the reference tree)."""
from oracle import twin as T
from oracle.sesdiff_twin import Confusable, script_to_str, shortest_edit_script

A = T.TEST_ALPHABET


def test0501_confusable_found_in():
    c = Confusable("-[y]+[i]", 1.1)
    assert c.found_in(shortest_edit_script("huys", "huis"))
    assert not c.found_in(shortest_edit_script("huys", "huls"))


def _model(script):
    m = T.VariantModel(A)
    for w in ("huis", "huls"):
        m.add_to_vocabulary(w)
    m.add_to_confusables(script, 1.1)
    m.build()
    return m


def test0502_0503_confusable_boosts_huis():
    m = _model("-[y]+[i]")
    for q in ("huys", "Huys"):
        r = m.find_variants(q, T.test_searchparams())
        assert [m.decoder[x.vocab_id].text for x in r] == ["huis", "huls"]
        assert r[0].dist_score > r[1].dist_score


def test0504_confusable_nomatch():
    m = _model("-[y]+[p]")
    r = m.find_variants("Huys", T.test_searchparams())
    assert len(r) == 2 and r[0].dist_score == r[1].dist_score


def test_edit_scripts_known_shapes():
    """diff-match-patch as published: prefix/suffix trimming, bisect, semantic clean-up (kitten/sitting is its
    documented example: the one-letter equality between two edits is absorbed), edge preference of lossless shifts."""
    assert script_to_str(shortest_edit_script("huys", "huis")) == "=[hu]-[y]+[i]=[s]"
    assert script_to_str(shortest_edit_script("separate", "seperate")) == "=[sep]-[a]+[e]=[rate]"
    assert script_to_str(shortest_edit_script("kitten", "sitting")) == "-[k]+[s]=[itt]-[en]+[ing]"
    assert script_to_str(shortest_edit_script("aab", "ab")) == "-[a]=[ab]"
    assert script_to_str(shortest_edit_script("", "abc")) == "+[abc]"
    assert script_to_str(shortest_edit_script("abc", "abc")) == "=[abc]"


def test_pattern_syntax():
    c = Confusable("^=[c|k]-[y]+[i]$", 0.9)
    assert c.strictbegin and c.strictend and c.instructions == [("=", ["c", "k"]), ("-", ["y"]), ("+", ["i"])]
    assert Confusable("=[c|k]-[y]+[i]", 1.1).found_in(shortest_edit_script("cylinder", "cilinder"))
    assert not Confusable("=[c|k]-[y]+[i]", 1.1).found_in(shortest_edit_script("huys", "huis"))
    assert Confusable("-[y]+[i]$", 1.1).found_in(shortest_edit_script("hay", "hai"))
    assert not Confusable("-[y]+[i]$", 1.1).found_in(shortest_edit_script("huys", "huis"))


def test_product_edit_scripts_equal_twin(data_dir):
    """The product's host-side diff (csrc/confusables.cpp, anx_edit_script: no GPU involved) against the twin's on
    4000 (misspelling, word) pairs of the eng lexicon plus hand-picked shapes.  Both restate the same published
    algorithm independently (C++ / Python); this pins them to each other, not to the reference (see module doc)."""
    import os
    import random
    import analiticcl_amd as A
    from analiticcl_amd import synth
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    rng = random.Random(5)
    qs = synth.make_queries(words, 4000, max_len=20, seed=9)
    pairs = [(q, rng.choice(words)) if i % 4 == 0 else (q, min(rng.sample(words, 40), key=lambda w: abs(len(w) - len(q)) + sum(a != b for a, b in zip(w, q))))
             for i, q in enumerate(qs)]
    pairs += [("kitten", "sitting"), ("aab", "ab"), ("", "abc"), ("abc", ""), ("abc", "abc"), ("abXcd", "efXgh"),
              ("the cat sat", "the hat sits"), ("étude", "etude"), ("naïve", "naive"), ("abcabcabc", "abXabXabc"),
              ("mississippi", "misisipi"), ("a b  c", "a  b c")]
    for a, b in pairs:
        assert A.edit_script(a, b) == script_to_str(shortest_edit_script(a, b)), (a, b)


def test_product_pattern_errors():
    import pytest
    import analiticcl_amd as A
    m = A.VariantModel("", alphabet_text="a\nb\n")
    with pytest.raises(A.AnxError):
        m.add_to_confusables("y>i", 1.1)
    with pytest.raises(A.AnxError):
        m.add_to_confusables("-[]", 1.1)
    m.add_to_confusables("^=[c|k]-[y]+[i]$", 0.9)


def test_composed_confusable_oracle_equals_twin(data_dir):
    """oracle/confusable_oracle.py (C oracle up to the crop + sesdiff twin rescoring + re-rank + cutoff) against the full
    pure-Python twin on a 3000-word nld subset with the committed 10-pattern list: the composition the full-size GPU test
    of BASELINE configs[2] uses as its checker (the twin alone needs seconds per query on the whole lexicon)."""
    import os
    from analiticcl_amd import synth
    from oracle import cwrap as O
    from oracle import confusable_oracle as CO
    alpha = os.path.join(data_dir, "simple.alphabet.tsv")
    conf_path = os.path.join(synth.GOLDEN_DATA, "confusables10.tsv")
    allw = synth.load_lexicon_words(os.path.join(data_dir, "nld.aspell.lexicon"))
    # contiguous (alphabetical) slices: neighbouring word forms, so that queries have several near candidates
    words = allw[30000:31500] + allw[120000:121500] + [w for w in allw if "y" in w or "ck" in w][:600]
    tw = T.VariantModel(T.read_alphabet(alpha))
    o = O.OracleModel(alphabet_path=alpha)
    for w in words:
        tw.add_to_vocabulary(w)
        o.add(w)
    tw.read_confusablelist(conf_path)
    tw.build()
    o.build()
    confs = CO.read_confusables(conf_path)
    assert len(confs) == 10
    qs = synth.make_queries(words, 250, max_len=24, seed=31)
    fired = 0
    for mm, cut in ((10, 2.0), (3, 1.2)):
        tp = T.SearchParameters(("abs", 3), ("abs", 3), mm, 0.25, cut, False, 0.0)
        op = O.make_params(("abs", 3), ("abs", 3), mm, 0.25, 0.0)  # no cutoff yet: it follows the late rescoring
        for q in qs:
            exp = [(x.vocab_id, x.dist_score, x.freq_score) for x in tw.find_variants(q, tp)]
            got = CO.late_rescore(o.find_variants(q, op), q, confs, o.text, 0.0, cut)
            assert got == exp, (q, got[:3], exp[:3])
            fired += any(CO.confusable_weight(confs, q, o.text(v)) != 1.0 for v, _d, _f in got)
    assert fired > 20


def test_product_confusable_weight_equals_twin(data_dir):
    """anx_model_confusable_weight (host: screen + edit script + pattern matcher) against the twin's compute_confusable_weight
    on 6000 (misspelling, word) pairs and patterns of every shape: plain, multi-character, alternatives, context, `^`, `$`,
    single-instruction equalities.  The screen that skips the edit script when no pattern can match must never change a weight."""
    import os
    import random
    import analiticcl_amd as A
    from analiticcl_amd import synth
    pats = [("-[y]+[i]", 1.1), ("-[i]+[y]", 1.2), ("-[ck]+[k]", 1.05), ("-[c]+[k]", 1.3), ("-[ae]+[e]", 1.15), ("-[s]+[z]", 0.9),
            ("=[c|k]-[y]+[i]", 1.07), ("+[e]$", 0.95), ("^-[h]", 0.8), ("-[e]$", 0.85), ("^+[s]", 1.25), ("=[t]+[t]", 1.4),
            ("-[e]=[r]", 0.7), ("+[s]", 0.97), ("=[ing]$", 1.01), ("^=[un]", 1.02), ("-[a|e|i]+[o|u]", 1.11), ("+[é]", 1.5),
            ("=[a]", 1.001), ("^=[b]-[a]", 1.6)]
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))[::29][:4000] + ["café", "cafe", "naïve"]
    tw = T.VariantModel(T.read_alphabet(os.path.join(data_dir, "simple.alphabet.tsv")))
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights())
    ids = []
    for w in words:
        ids.append(tw.add_to_vocabulary(w))
        assert g.add_to_vocabulary(w) == ids[-1]
    for pat, wt in pats:
        tw.add_to_confusables(pat, wt)
        g.add_to_confusables(pat, wt)
    rng = random.Random(13)
    qs = synth.make_queries(words, 6000, max_len=20, seed=17)
    fired = 0
    for q in qs:
        near = min(rng.sample(range(len(words)), 30), key=lambda i: abs(len(words[i]) - len(q)) + sum(a != b for a, b in zip(words[i], q)))
        for i in (near, rng.randrange(len(words))):
            exp = tw.compute_confusable_weight(q, ids[i])
            assert g.compute_confusable_weight(q, ids[i]) == exp, (q, words[i])
            fired += exp != 1.0
    assert fired > 1000
