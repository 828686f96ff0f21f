"""GPU parity for search mode (anx_find_all_matches_batch): the product's C++ search
driver over the HIP variant-query path against (1) the reference's own 07xx tests (/root/reference/tests/main.rs,
values transcribed), (2) the tutorial's recorded find_all_matches outputs and (3) the oracle twin on random texts.

For (3) the twin's segmentation / lattice / LM code runs unchanged; only its per-segment find_variants is served by the
C oracle (same results as the twin, tests/test_oracle_c.py) so that hundreds of segments finish in seconds."""
import os
import random

import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import cwrap as O
from oracle import twin as T

from search_common import TwinOverOracle

TEST_ALPHABET_TSV = "\n".join(f"{c}\t{c.upper()}" for c in "abcdefghijklmnopqrstuvwxyz") + "\n.\t,\n"
LM = A.VocabParams(vocabtype="LM")


def sparams(**kw):  # src/test.rs:48-68
    d = dict(max_anagram_distance=2, max_edit_distance=2, max_matches=10, score_threshold=0.0, cutoff_threshold=0.0,
             max_ngram=2)
    d.update(kw)
    return A.SearchParameters(**d)


def small(words, lm=(), freq=None):
    g = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=0)
    for w in words:
        g.add_to_vocabulary(w, freq)
    for t, f in lm:
        g.add_to_vocabulary(t, f, LM)
    g.build()
    return g


def lm_model(extra_words=(), extra_lm=()):
    return small(("I", "think", "sink", "you", "are", "right") + tuple(extra_words) + ("are right",),
                 (("<bos> I", 2), ("I think", 2), ("I sink", 1), ("you are", 2), ("right <eos>", 2)) + tuple(extra_lm), 2)


def best(matches):
    return [(m["input"], m["variants"][0]["text"] if m["variants"] else m["input"]) for m in matches]


def test0701_unigram_only():  # tests/main.rs:1121-1141
    g = small(("I", "think", "sink", "you", "are", "right"))
    r = g.find_all_matches("I tink you are rihgt", sparams(max_ngram=1))
    assert best(r) == [("I", "I"), ("tink", "think"), ("you", "you"), ("are", "are"), ("rihgt", "right")]


def test0702_0705_lm():  # tests/main.rs:1144-1424
    g = lm_model()
    exp = [("I", "I"), ("tink", "think"), ("you", "you"), ("are rihgt", "are right")]
    r = g.find_all_matches("I tink you are rihgt", sparams())
    assert best(r) == exp and (r[1]["offset"]["begin"], r[1]["offset"]["end"]) == (2, 6)
    r = g.find_all_matches("I tink you are\nrihgt", sparams())
    assert best(r) == exp[:3] + [("are\nrihgt", "are right")]
    assert best(g.find_all_matches("I tink you are rihgt", sparams(lm_weight=0.0))) == exp
    g2 = lm_model(("am", "sure"), (("I am", 2), ("sure <eos>", 2)))
    r = g2.find_all_matches("I tink you are rihgt\n\nI am sur", sparams())
    assert best(r) == exp + [("I", "I"), ("am", "am"), ("sur", "sure")]


def test0706_0707_offsets():  # tests/main.rs:1427-1481
    g = small(("I", "think", "you", "are", "right"))
    r = g.find_all_matches("I thиnk you are righт", sparams(max_ngram=1, unicodeoffsets=True))
    assert [m["input"] for m in r] == ["I", "thиnk", "you", "are", "righт"]
    assert (r[1]["offset"]["begin"], r[1]["offset"]["end"]) == (2, 7)
    assert best(r)[1][1] == "think" and best(r)[4][1] == "right"
    r = g.find_all_matches("I thиnk you are rihgt", sparams(max_ngram=1))
    assert (r[1]["offset"]["begin"], r[1]["offset"]["end"]) == (2, 8) and best(r)[4][1] == "right"


def test_empty_and_unbuilt():
    g = small(("a",))
    assert g.find_all_matches("", sparams()) == []
    assert g.find_all_matches("   ", sparams()) == []
    assert g.find_all_matches_ids([], sparams()) == []


@pytest.fixture(scope="module")
def eng(data_dir):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    g.build()
    return g


def test_tutorial_find_all_matches(eng, tutorial_outputs):
    """tutorial.ipynb cells 22 and 24 (recorded outputs of the reference)."""
    p = A.SearchParameters(unicodeoffsets=True)
    case = tutorial_outputs["find_all_matches"][0]
    r = eng.find_all_matches(case["input"], p)
    assert [(m["input"], m["offset"]["begin"], m["offset"]["end"]) for m in r] == \
        [(c["input"], c["begin"], c["end"]) for c in case["matches"]]
    for m, c in zip(r, case["matches"]):
        assert [[v["text"], v["score"], v["dist_score"], v["freq_score"]] for v in m["variants"]] == c["variants"]
    case = tutorial_outputs["find_all_matches"][1]
    m = eng.find_all_matches(case["input"], p)[case["only_match_index"]]
    c = case["matches"][0]
    assert (m["input"], m["offset"]["begin"], m["offset"]["end"]) == (c["input"], c["begin"], c["end"])
    assert [[v["text"], v["score"], v["dist_score"], v["freq_score"]] for v in m["variants"]] == c["variants"]


def random_texts(words, phrases, n, seed):
    rng = random.Random(seed)
    qs = synth.make_queries(words, n * 9, max_len=14, seed=seed)
    ps = synth.make_queries(phrases, n * 3, max_len=24, seed=seed + 1)
    for i in range(0, len(qs), 4):
        qs[i] = ps[i // 4 % len(ps)]
    seps = [" "] * 12 + [", ", ". ", "\n", "-", "'", "; ", " (", ") ", "\n\n", "  ", " é ", ": \""]
    texts, k = [], 0
    for _ in range(n):
        nw = rng.randrange(1, 9)
        t = ""
        for j in range(nw):
            t += qs[k]
            k += 1
            if j + 1 < nw or rng.random() < 0.3:
                t += rng.choice(seps)
        texts.append(t)
    return texts


def build_world(with_lm, with_rules):
    """3000 lexicon words + 300 indexed phrases (+ a bigram LM over the frequent words, + 300 context rules): the same model as
    product (device), twin and oracle."""
    words = [w for w in synth.load_lexicon_words(os.path.join(synth.GOLDEN_DATA, "eng_aspell.lexicon.gz")) if w.isascii() and w.isalpha()][::37][:3000]
    rng = random.Random(11)
    tw = TwinOverOracle(T.TEST_ALPHABET)
    orc = O.OracleModel(alphabet_text=TEST_ALPHABET_TSV)
    g = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=0)
    for w in words:
        f = rng.randrange(1, 50)
        tw.add_to_vocabulary(w, f)
        orc.add(w, f)
        g.add_to_vocabulary(w, f)
    phrases = []
    for _ in range(300):  # indexed multi-word entries, so that bigram / trigram segments do find variants
        ph = " ".join(rng.choice(words[:400]) for _ in range(rng.choice((2, 2, 3))))
        if len(ph) <= 22 and ph not in phrases:
            phrases.append(ph)
            f = rng.randrange(1, 50)
            tw.add_to_vocabulary(ph, f)
            orc.add(ph, f)
            g.add_to_vocabulary(ph, f)
    if with_lm:
        for _ in range(4000):
            a, b = rng.choice(words[:400]), rng.choice(words[:400])
            f = rng.randrange(1, 9)
            tw.add_lm(f"{a} {b}", f)
            g.add_to_vocabulary(f"{a} {b}", f, LM)
        for w in words[:50]:
            tw.add_lm(f"<bos> {w}", 3)
            g.add_to_vocabulary(f"<bos> {w}", 3, LM)
    tw.build()
    orc.build()
    g.build()
    tw.attach(orc)
    if with_rules:  # context rules over frequent words: bonus / penalty, wildcards, negation, disjunction, tags
        common = words[:400]
        for i in range(300):
            kind = i % 6
            a, b, c = rng.choice(common), rng.choice(common), rng.choice(common)
            pattern = (a, f"{a}; ?", f"^; {a}", f"{a}|{b}; !{c}", f"!({a}|{b}); {c}", f"?; {a}; ^")[kind]
            score = rng.choice((0.5, 0.8, 0.9, 1.1, 1.3, 2.0))
            tags, offs = ((), ()) if i % 3 == 0 else ((f"t{i % 7}",), ()) if i % 3 == 1 else ((f"t{i % 5}", "u"), ("0:1", ":"))
            tw.add_contextrule(pattern, score, list(tags), list(offs))
            g.add_contextrule(pattern, score, list(tags), list(offs))
        assert g.tags == tw.tags
    return g, tw, words, phrases


def compare_with_twin(g, tw, texts, max_seq):
    """-> (matches with n > 1, tagged matches); every Match field of the product equals the twin's"""
    gp = A.SearchParameters(max_anagram_distance=2, max_edit_distance=2, max_matches=6, score_threshold=0.3,
                            cutoff_threshold=0.0, max_ngram=3, max_seq=max_seq)
    tp = T.SearchParams(("abs", 2), ("abs", 2), 6, 0.3, 0.0, False, 0.0, max_ngram=3, max_seq=max_seq)
    got = g.find_all_matches_ids(texts, gp)
    n_multi = n_tagged = 0
    for text, gm in zip(texts, got):
        exp = tw.find_all_matches(text, tp)
        raw = text.encode()
        assert [(raw[m["begin"]:m["end"]].decode(), m["begin"], m["end"]) for m in gm] == \
            [(e.text, e.begin, e.end) for e in exp], text
        for m, e in zip(gm, exp):
            ev = e.variants or []
            assert [v[0] for v in m["variants"]] == [v.vocab_id for v in ev], (text, e.text)
            for v, w in zip(m["variants"], ev):
                assert abs(v[1] - w.dist_score) < 1e-6 and abs(v[2] - w.freq_score) < 1e-6
            if ev:
                assert m["selected"] == e.selected, (text, e.text)
            assert (m["tag"], m["seqnr"]) == (e.tag, e.seqnr), (text, e.text)
            n_multi += e.n > 1
            n_tagged += bool(e.tag)
    return n_multi, n_tagged


@pytest.mark.parametrize("with_lm,with_rules", [(False, False), (True, False), (False, True), (True, True)])
def test_random_texts_vs_twin(with_lm, with_rules):
    """Markov-free small world: 3000 lexicon words (+ a bigram LM over them when with_lm), 200 random texts, the whole
    batch in ONE anx_find_all_matches_batch call; every Match field must equal the twin's."""
    g, tw, words, phrases = build_world(with_lm, with_rules)
    texts = random_texts(words[:400] if with_lm else words, phrases, 200, 5 + int(with_lm))
    n_multi, n_tagged = compare_with_twin(g, tw, texts, 40)
    assert n_multi > 0  # the lattice did pick some bigram/trigram segments
    assert (n_tagged > 0) == with_rules


@pytest.mark.parametrize("with_lm", [False, True])
def test_long_stretches_vs_twin(with_lm):
    """Stretches without a hard boundary (words separated by one space) of 60, 700 and 2300 words: long lattices through the
    k-best merge, the per-node LM sums and the back-pointer walk; a small max_seq on all of them and the default 250 on the
    first."""
    g, tw, words, phrases = build_world(with_lm, False)
    pool = synth.make_queries(words[:400], 3200, max_len=14, seed=21)
    ph = synth.make_queries(phrases, 200, max_len=24, seed=22)
    for i in range(0, len(pool), 9):
        pool[i] = ph[i // 9 % len(ph)]
    texts = [" ".join(pool[:60]), " ".join(pool[60:760]) + ".", " ".join(pool[760:3060])]
    n_multi, _ = compare_with_twin(g, tw, texts, 6)
    assert n_multi > 0
    compare_with_twin(g, tw, texts[:1], 250)


# -- context rules: tests/main.rs:1575-1800 (values transcribed) ----------------------------------------------------
def rules_model():
    return small(("I", "think", "sink", "you", "are", "right"), freq=2)


def chosen(r):
    return [m["variants"][0]["text"] if m["variants"] else m["input"] for m in r]


def test0902_0905_context_rules():
    p = sparams(max_ngram=1, lm_weight=0.0)
    text = "I tink you are rihgt"
    g = rules_model()
    g.add_contextrule("I; think", 1.1, ["testtag"], [])  # bonus
    r = g.find_all_matches(text, p)
    assert chosen(r) == ["I", "think", "you", "are", "right"]
    assert (r[0]["tag"], r[0]["seqnr"], r[1]["tag"], r[1]["seqnr"]) == (["testtag"], [0], ["testtag"], [1])
    assert "tag" not in r[2]
    g = rules_model()
    g.add_contextrule("I; think", 0.9)  # penalty
    assert chosen(g.find_all_matches(text, p)) == ["I", "sink", "you", "are", "right"]
    g = rules_model()
    for w in ("think", "are", "right"):
        g.add_contextrule(w, 1.0, ["testtag"])
    r = g.find_all_matches(text, p)
    assert chosen(r) == ["I", "think", "you", "are", "right"]
    assert [m.get("tag", []) for m in r] == [[], ["testtag"], [], ["testtag"], ["testtag"]]
    assert [m.get("seqnr", []) for m in r] == [[], [0], [], [0], [0]]
    g = rules_model()
    g.add_contextrule("I; think", 1.1, ["testtag", "testtag2"])
    r = g.find_all_matches(text, p)
    assert (r[0]["tag"], r[0]["seqnr"], r[1]["tag"], r[1]["seqnr"]) == \
        (["testtag", "testtag2"], [0, 0], ["testtag", "testtag2"], [1, 1])
    assert g.tags == ["testtag", "testtag2"]


def test_contextrules_file_and_lexicons(tmp_path):
    """read_contextrules (src/lib.rs:570-656) with @lexicon patterns over two lexicon files: product == twin."""
    amph, rept = tmp_path / "amphibians.tsv", tmp_path / "reptiles.tsv"
    amph.write_text("salamander\t3\nfrog\t3\ntoad\t2\nnewt\t1\n")
    rept.write_text("lizard\t3\nsnake\t3\nskink\t1\nnewt\t1\n")
    rules = tmp_path / "rules.tsv"
    rules.write_text("# pattern\tscore\ttags\toffsets\n"
                     "@amphibians.tsv; @reptiles.tsv\t1.2\tpair\n"
                     "\n"
                     "@reptiles.tsv\t0.9\treptile; any\t0:1; :\n"
                     "^; frog\t1.1\n"
                     "!(snake|toad); newt|skink\t1.05\tx\t1:1\n")
    g = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=0)
    tw = T.SearchModel(T.TEST_ALPHABET)
    for f in (amph, rept):
        g.read_lexicon(str(f))
        tw.read_vocabulary(str(f))
    g.build()
    tw.build()
    g.read_contextrules(str(rules))
    tw.read_contextrules(str(rules))
    assert g.tags == tw.tags == ["pair", "reptile", "any", "x"]
    gp = sparams(max_ngram=1, lm_weight=0.0)
    tp = T.test_searchparams_search()
    tp.max_ngram, tp.lm_weight = 1, 0.0
    for text in ("salamnder lizrd frog snke toad", "qqqq frog newt skink", "snake newt lizard, toad skink!", "nwet"):
        got = g.find_all_matches(text, gp)
        exp = tw.find_all_matches(text, tp)
        assert [m["input"] for m in got] == [e.text for e in exp]
        assert chosen(got) == [tw.match_to_str(e) for e in exp], text
        assert [m.get("tag", []) for m in got] == [[tw.tags[t] for t in e.tag] for e in exp], text
        assert [m.get("seqnr", []) for m in got] == [e.seqnr for e in exp], text
    with pytest.raises(A.AnxError):
        g.add_contextrule("notaword", 1.0)
    with pytest.raises(A.AnxError):
        g.add_contextrule("@nolexicon.tsv", 1.0)
    with pytest.raises(A.AnxError):
        g.add_contextrule("frog", 1.0, ["t"], ["x:1"])
    bad = tmp_path / "bad.tsv"
    bad.write_text("frog\tnotanumber\n")
    with pytest.raises(A.AnxError):
        g.read_contextrules(str(bad))


def test_cli_query_readme_line(data_dir, tmp_path, capsys):
    """README.md:121-124 (recorded output of `analiticcl query` with the CLI defaults) through `python -m analiticcl_amd query`."""
    from analiticcl_amd import cli
    inp = tmp_path / "in.txt"
    inp.write_text("seperate\n", encoding="utf-8")
    assert cli.main(["query", "--lexicon", os.path.join(data_dir, "eng.aspell.lexicon"), "--alphabet",
                     os.path.join(data_dir, "simple.alphabet.tsv"), str(inp)]) == 0
    out = capsys.readouterr().out
    # README.md:121-124: same seven variants and scores.  The README prints the tied pair as `separates`, `separated`;
    # the reference's enumeration order (ascending anagram value, src/lib.rs:1148) and tutorial.ipynb's recorded
    # outputs ('separated' before 'separates' at equal score, cells 18/20) give the order asserted here.
    assert out == ("seperate\tseparate\t0.734375\t\toperate\t0.6875\t\tdesperate\t0.6875\t\ttemperate\t0.6875\t\tserrate\t0.65625\t"
                   "\tseparated\t0.609375\t\tseparates\t0.609375\t\n")
    assert cli.main(["search", "--lexicon", os.path.join(data_dir, "eng.aspell.lexicon"), "--alphabet",
                     os.path.join(data_dir, "simple.alphabet.tsv"), "--json", str(inp)]) == 0
    import json
    js = json.loads(capsys.readouterr().out)
    assert js[0]["input"] == "seperate" and js[0]["begin"] == 0 and js[0]["end"] == 8 and js[0]["variants"][0]["text"] == "separate"


def test_native_query_output_equals_python_formatter(data_dir, tmp_path):
    """anx_format_query_output (the text `query` mode prints) == tsv_line / json_item of analiticcl_amd/cli.py (which
    tests/test_cli_cpu.py pins to the reference's formats), incl. lexicon names, frequency-weighted scores, `via` of
    variant lists, quotes in the input and empty result lists."""
    from analiticcl_amd import cli
    lex2 = tmp_path / "extra.tsv"
    lex2.write_text("separate\t5\nzebra\t2\nquote\"d\t1\n")
    vl = tmp_path / "variants.tsv"
    vl.write_text("separate\tseperate\t1.0\tseparete\t0.9\n")
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    g.read_lexicon(str(lex2))
    g.read_variants(str(vl))
    g.build()
    qs = ["seperate", "zebr", 'quote"d', "", "xqzzyvw", "recieve", "separete", "tesst"] + \
        synth.make_queries(synth.load_lexicon_words(os.path.join(synth.GOLDEN_DATA, "eng_aspell.lexicon.gz")), 300, seed=3)
    for fw in (0.0, 0.5):
        p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25,
                               cutoff_threshold=2.0, freq_weight=fw)
        items = g.find_variants_par(qs, p)
        for lexmatch in (False, True):
            exp = "".join(cli.tsv_line(it["input"], it["variants"], None, lexmatch) + "\n" for it in items)
            assert g.query_output(qs, p, False, lexmatch) == exp
            exp = "".join(cli.json_item(it["input"], it["variants"], 5 + i, None, lexmatch) for i, it in enumerate(items))
            assert g.query_output(qs, p, True, lexmatch, 5) == exp
        first = g.query_output(qs[:2], p, True, False, 1)
        assert first.startswith('    { "input": "seperate"') and '\n    ,{ "input": "zebr"' in first
    assert any("via" in v for it in items for v in it["variants"])


def test_native_search_output_equals_python_formatter():
    """anx_format_search_output == tsv_line / json_item over find_all_matches (selected variant first, offsets, tags)."""
    from analiticcl_amd import cli
    g = lm_model()
    g.add_contextrule("I; think", 1.1, ["subj", "pair"], ["0:1", ":"])
    texts = ["I tink you are rihgt", "you are\nrihgt, I sink!", "", "zzzzqq I"]
    for kw in (dict(), dict(lm_weight=0.0), dict(max_ngram=1)):
        p = sparams(**kw)
        for lexmatch in (False, True):
            for js in (False, True):
                seq, exp = 1, ""
                for t in texts:
                    for m in (g.find_all_matches(t, p) if t else []):
                        off = (m["offset"]["begin"], m["offset"]["end"])
                        exp += (cli.json_item(m["input"], m["variants"], seq, off, lexmatch, m.get("tag", ()), m.get("seqnr", ()))
                                if js else cli.tsv_line(m["input"], m["variants"], off, lexmatch) + "\n")
                        seq += 1
                got, n = g.search_output(texts, p, js, lexmatch, 1)
                assert got == exp and n == seq - 1
    assert '"tag": ["subj","pair"], "seqnr": [ 0,0]' in g.search_output(["I tink"], sparams(), True)[0]
    with pytest.raises(ValueError):
        g.search_output(["I"], sparams(unicodeoffsets=True))


def test_cli_search_with_contextrules(tmp_path, capsys):
    """`search --contextrules FILE --json`: the rule file is read after the lexicons (bin:1097-1109) and the tags of the
    winning sequence are printed (bin:99-121)."""
    import json
    from analiticcl_amd import cli
    alphabet = tmp_path / "alphabet.tsv"
    alphabet.write_text(TEST_ALPHABET_TSV)
    lex = tmp_path / "lex.tsv"
    lex.write_text("".join(f"{w}\t2\n" for w in ("I", "think", "sink", "you", "are", "right")))
    rules = tmp_path / "rules.tsv"
    rules.write_text("I; think\t1.1\tsubject-verb\n")
    inp = tmp_path / "in.txt"
    inp.write_text("I tink you are rihgt\n")
    argv = ["search", "--lexicon", str(lex), "--alphabet", str(alphabet), "--json", "-k", "2", "-d", "2", "--weight-lm", "0",
            "--max-ngram-order", "1", str(inp)]
    assert cli.main(argv + ["--contextrules", str(rules)]) == 0
    js = json.loads(capsys.readouterr().out)
    assert [m["variants"][0]["text"] for m in js] == ["I", "think", "you", "are", "right"]
    assert (js[0]["tag"], js[0]["seqnr"], js[1]["tag"], js[1]["seqnr"]) == (["subject-verb"], [0], ["subject-verb"], [1])
    assert "tag" not in js[2]
    rules.write_text("I; think\t0.9\n")  # the penalty makes "sink" win (tests/main.rs:1613-1640)
    assert cli.main(argv + ["-R", str(rules)]) == 0
    js = json.loads(capsys.readouterr().out)
    assert [m["variants"][0]["text"] for m in js] == ["I", "sink", "you", "are", "right"]


def test_cli_index_cache_roundtrip(data_dir, tmp_path, capsys):
    """--index-cache: the first run builds the model and writes the image, the second one loads it (no lexicon read, no
    index build) and must print the same bytes -- query and search mode, with a context-rule file on top of the image."""
    from analiticcl_amd import cli
    cache = tmp_path / "model.idx"
    inp = tmp_path / "in.txt"
    inp.write_text("seperate\nrecieve teh mesage\nacommodate\n", encoding="utf-8")
    rules = tmp_path / "rules.tsv"
    rules.write_text("receive; the\t1.1\tpair\n")
    base = ["--lexicon", os.path.join(data_dir, "eng.aspell.lexicon"), "--alphabet", os.path.join(data_dir, "simple.alphabet.tsv"),
            "--index-cache", str(cache)]
    outs = []
    for _ in range(2):
        assert cli.main(["query"] + base + [str(inp)]) == 0
        q = capsys.readouterr().out
        assert cli.main(["search"] + base + ["--json", "--contextrules", str(rules), str(inp)]) == 0
        outs.append((q, capsys.readouterr().out))
        assert cache.exists()
    assert outs[0] == outs[1]
    assert outs[0][0].startswith("seperate\tseparate\t0.734375\t") and '"input": "recieve"' in outs[0][1]
    # a stale image must not be used: the same cache file with an EDITED lexicon (the misspelling itself added) is rebuilt
    lex2 = tmp_path / "edited.lexicon"
    with open(os.path.join(data_dir, "eng.aspell.lexicon"), encoding="utf-8") as f:
        lex2.write_text(f.read() + "seperate\n", encoding="utf-8")
    base2 = ["--lexicon", str(lex2), "--alphabet", os.path.join(data_dir, "simple.alphabet.tsv"), "--index-cache", str(cache)]
    tag_before = A.VariantModel.index_tag_of(str(cache))
    assert cli.main(["query"] + base2 + [str(inp)]) == 0
    assert capsys.readouterr().out.startswith("seperate\tseperate\t1\t")
    assert A.VariantModel.index_tag_of(str(cache)) != tag_before          # rewritten for the new resource list
    assert cli.main(["query"] + base + [str(inp)]) == 0                   # and back: rebuilt again, not served from the edited image
    assert capsys.readouterr().out == outs[0][0]
    assert A.VariantModel.index_tag_of(str(tmp_path / "nonexistent.idx")) is None


def test_python_binding_multiple_lexicons(data_dir, tmp_path):
    """The reference's only Python-binding test (/root/reference/bindings/python/tests/tests.py:12-36, values and the two
    word lists transcribed): two lexicons, find_all_matches with max_edit_distance=3 / max_ngram=1, every match's best
    variant and the `lexicons` attribution of that variant."""
    amph = tmp_path / "amphibians.tsv"
    amph.write_text("axolotl\nfrog\nnewt\nsalamander\ntoad", encoding="utf-8")   # no trailing newline, as in the reference file
    rept = tmp_path / "reptiles.tsv"
    rept.write_text("iguana\nlizard\nsnake\nturtle", encoding="utf-8")
    model = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), debug=False)
    model.read_lexicon(str(amph))
    model.read_lexicon(str(rept))
    model.build()
    results = model.find_all_matches("Salamander lizard frog snake toad", A.SearchParameters(max_edit_distance=3, max_ngram=1))
    assert len(results) == 5
    for result, orig_term, lexicon, lex_term in ((results[0], "Salamander", amph, "salamander"), (results[1], "lizard", rept, "lizard"),
                                                 (results[2], "frog", amph, "frog"), (results[3], "snake", rept, "snake"),
                                                 (results[4], "toad", amph, "toad")):
        assert result["input"] == orig_term
        assert len(result["variants"]) > 0
        best_match = result["variants"][0]
        assert best_match["text"] == lex_term
        assert best_match["lexicons"] == [str(lexicon)]


def test_device_lattice_equals_host_lattice(data_dir):
    """Search mode's lattice decoding on the device (lattice.hip: k best paths per state by a wave-wide merge, LM sums per lattice
    node, portable_log normalisation) against the host decoder (ANX_LATTICE=host): every match field of every text identical, with
    and without a language model, for several max_seq / max_ngram / weight settings."""
    import random
    import numpy as np
    lex = os.path.join(data_dir, "eng.aspell.lexicon")
    words = synth.load_lexicon_words(lex)
    rng = random.Random(7)
    common = [w for w in words if w.isalpha()][::23][:5000]
    LM = A.VocabParams(vocabtype="LM")
    for with_lm in (True, False):
        g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
        g.read_lexicon(lex)
        if with_lm:
            for _ in range(20000):
                g.add_to_vocabulary(f"{rng.choice(common)} {rng.choice(common)}", rng.randrange(1, 20), LM)
            for w in common[:500]:
                g.add_to_vocabulary(f"<bos> {w}", 5, LM)
            for w in common[:300]:
                g.add_to_vocabulary(w, rng.randrange(1, 50), LM)
        g.build()
        texts = synth.make_running_text(common, 1.2, seed=31) + ["", "one", "it's a well-known co-op, isn't it?", "a " * 300, "zzqx " * 40 + "end."]
        for kw in (dict(max_ngram=3), dict(max_ngram=2, max_seq=7), dict(max_ngram=3, max_seq=1), dict(max_ngram=3, lm_weight=0.0),
                   dict(max_ngram=1), dict(max_ngram=3, max_matches=20, variantmodel_weight=1.0, lm_weight=2.0),
                   dict(max_ngram=3, max_seq=5000)):   # beyond the device's node pools: handed back to the host decoder
            p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, **({"max_matches": 10} | kw))
            tx = texts if kw.get("max_seq", 250) <= 4096 else texts[:300] + texts[-5:]
            out = {}
            for mode in ("device", "host"):
                A.set_switch("ANX_LATTICE", "host" if mode == "host" else None)
                try:
                    out[mode] = g.find_all_matches_arrays(tx, p)
                finally:
                    A.set_switch("ANX_LATTICE", None)
            for x, y in zip(out["device"], out["host"]):
                assert np.array_equal(x, y), (with_lm, kw)
            assert out["device"][1].size > (50_000 if tx is texts else 5_000)


@pytest.mark.gpu
def test_search_parts_equal_one_pass(data_dir):
    """A large find_all_matches call runs as concurrent parts (ANX_SEARCH_PARTS, default 2: one part's device work under the other's
    host phases) whose arrays are merged: offsets, every match field, every variant row and the tags identical to one pass."""
    import random
    import numpy as np
    lex = os.path.join(data_dir, "eng.aspell.lexicon")
    words = synth.load_lexicon_words(lex)
    rng = random.Random(11)
    common = [w for w in words if w.isalpha()][::23][:5000]
    LM = A.VocabParams(vocabtype="LM")
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(lex)
    for _ in range(5000):
        g.add_to_vocabulary(f"{rng.choice(common)} {rng.choice(common)}", rng.randrange(1, 20), LM)
    g.add_contextrule("the; ?", 1.5, ["det", "noun"])
    g.build()
    texts = synth.make_running_text(common, 0.8, seed=5) + ["", "one", "the cat and the dgo", ""]
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, max_ngram=3)
    A.set_switch("ANX_SEARCH_PARTS_MIN", "1")
    try:
        ref = None
        for parts in ("1", "2", "3", "7"):
            A.set_switch("ANX_SEARCH_PARTS", parts)
            off, ma, ra = g.find_all_matches_arrays(texts, p)
            ids = g.find_all_matches_ids(texts[:200] + texts[-4:], p)   # the form with tags
            if ref is None:
                ref = (off, ma, ra, ids)
                assert off[-1] > 1000
                continue
            assert np.array_equal(off, ref[0]) and np.array_equal(ma, ref[1]) and np.array_equal(ra, ref[2]), parts
            assert ids == ref[3], parts
            # (the array form has no tag array: its output is written while later parts are still at work; written at the end instead:)
            A.set_switch("ANX_SEARCH_EARLY_OUTPUT", "0")
            off2, ma2, ra2 = g.find_all_matches_arrays(texts, p)
            A.set_switch("ANX_SEARCH_EARLY_OUTPUT", None)
            assert np.array_equal(off2, ref[0]) and np.array_equal(ma2, ref[1]) and np.array_equal(ra2, ref[2]), parts
            # unlimited lists (max_matches = 0: no bound for the rows, the output waits for the last part)
            if parts == "3":
                p0 = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=0, max_ngram=2)
                a3 = g.find_all_matches_arrays(texts[:300], p0)
                A.set_switch("ANX_SEARCH_PARTS", "1")
                a1 = g.find_all_matches_arrays(texts[:300], p0)
                for x, y in zip(a3, a1):
                    assert np.array_equal(x, y)
    finally:
        A.set_switch("ANX_SEARCH_EARLY_OUTPUT", None)
        A.set_switch("ANX_SEARCH_PARTS", None)
        A.set_switch("ANX_SEARCH_PARTS_MIN", None)


@pytest.mark.gpu
def test_one_device_pass_equals_the_classic_path(data_dir):
    """Search mode in one device pass (round 5: the lattices are built on the device from the batches' device-resident rows, the
    redundancy rule of /root/reference/src/search.rs:317-336 is applied there, only the chosen paths' matches come back) against the
    path of rounds 1-4 (ANX_SEARCH_ONEPASS=0: every ranked row downloaded, lattice input built by the host threads): offsets, every
    match field and every variant row identical -- on noisy text, on CLEAN text (nearly every higher-order segment redundant: their
    queries are cleared before the second batch runs), with a language model, frequency weighting and confusable patterns."""
    import random
    import numpy as np
    lex = os.path.join(data_dir, "eng.aspell.lexicon")
    words = synth.load_lexicon_words(lex)
    rng = random.Random(23)
    common = [w for w in words if w.isalpha()][::23][:5000]
    LM = A.VocabParams(vocabtype="LM")
    for variant in ("lm", "plain", "confusables"):
        g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
        g.read_lexicon(lex)
        if variant == "lm":
            for _ in range(8000):
                g.add_to_vocabulary(f"{rng.choice(common)} {rng.choice(common)}", rng.randrange(1, 20), LM)
            for w in common[:300]:
                g.add_to_vocabulary(f"<bos> {w}", 5, LM)
        if variant == "confusables":
            g.add_to_confusables("-[e]+[a]", 1.1)
            g.add_to_confusables("-[y]+[i]", 0.9)
        g.build()
        noisy = synth.make_running_text(common, 0.6, seed=41)
        clean = [" ".join(rng.choice(common) for _ in range(rng.randrange(3, 30))) + rng.choice([". ", "\n", ""]) for _ in range(1500)]
        texts = noisy + clean + ["", "x", "it's a well-known co-op", "a " * 200, "end"]
        for kw in (dict(max_ngram=3), dict(max_ngram=2, max_seq=3, freq_weight=0.5), dict(max_ngram=3, max_matches=0), dict(max_ngram=4, max_matches=3, lm_weight=0.0)):
            p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, **({"max_matches": 10} | kw))
            out = {}
            for mode in ("onepass", "classic"):
                A.set_switch("ANX_SEARCH_ONEPASS", "0" if mode == "classic" else None)
                try:
                    out[mode] = g.find_all_matches_arrays(texts, p)
                finally:
                    A.set_switch("ANX_SEARCH_ONEPASS", None)
            for x, y in zip(out["onepass"], out["classic"]):
                assert np.array_equal(x, y), (variant, kw)
            assert out["onepass"][1].size > 20_000


@pytest.mark.gpu
def test_lattice_kernel_equals_the_host_decoder_for_every_k(data_dir):
    """k_lattice's K best paths into a state (rustfst shortest_path(nshortest = max_seq), /root/reference/src/lib.rs:2288-2317) against
    the host decoder (ANX_LATTICE=host): every output array identical, for K below / at / above the number of candidate paths and
    around the number of list costs a head keeps in registers (max_seq 1, 2, 3, 7, 8, 9, 10, 17, 64, 250, 1000), with and without a
    language model, on text with long and short stretches."""
    import random
    import numpy as np
    lex = os.path.join(data_dir, "eng.aspell.lexicon")
    words = synth.load_lexicon_words(lex)
    rng = random.Random(5)
    common = [w for w in words if w.isalpha()][::23][:5000]
    LM = A.VocabParams(vocabtype="LM")
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(lex)
    for _ in range(8000):
        g.add_to_vocabulary(f"{rng.choice(common)} {rng.choice(common)}", rng.randrange(1, 20), LM)
    for w in common[:300]:
        g.add_to_vocabulary(f"<bos> {w}", 5, LM)
    g.build()
    noisy = synth.make_running_text(common, 0.6, seed=43)
    long_ones = [" ".join(rng.choice(common) for _ in range(rng.randrange(40, 120))) for _ in range(60)]   # stretches of up to 120 tokens
    texts = noisy + long_ones + ["", "x", "a b", "a " * 300, "it's a well-known co-op"]
    total = 0
    for kw in (dict(max_seq=1), dict(max_seq=2), dict(max_seq=3), dict(max_seq=7, max_ngram=2), dict(max_seq=8), dict(max_seq=9), dict(max_seq=10),
               dict(max_seq=17, max_ngram=2), dict(max_seq=64), dict(max_seq=250), dict(max_seq=250, lm_weight=0.0),
               dict(max_seq=1000, max_matches=3), dict(max_seq=250, max_ngram=4, max_matches=20)):
        p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, **({"max_matches": 10, "max_ngram": 3} | kw))
        out = {}
        for mode in ("device", "host"):
            if mode == "host":
                A.set_switch("ANX_LATTICE", "host")
            try:
                out[mode] = g.find_all_matches_arrays(texts, p)
            finally:
                A.set_switch("ANX_LATTICE", None)
        for x, y in zip(out["device"], out["host"]):
            assert np.array_equal(x, y), kw
        total += out["device"][1].size
    assert total > 100_000
