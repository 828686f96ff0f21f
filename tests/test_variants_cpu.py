"""Variant lists / transparent entries: oracle (twin + C) pinned by the reference's
test0801 (/root/reference/tests/main.rs:1484-1510) and tutorial.ipynb cells 27-32; product host logic on CPU."""
import random

import analiticcl_amd as A
from oracle import cwrap as O
from oracle import twin as T

TEST_ALPHABET_TSV = "\n".join(f"{c}\t{c.upper()}" for c in "abcdefghijklmnopqrstuvwxyz") + "\n.\t,\n"


def make_variant_file(path, rng, with_freq):
    words = ["house", "mouse", "horse", "hose", "moose", "louse", "hours", "our", "use", "muse", "rouse", "shout"]
    lines = []
    for w in rng.sample(words, 7):
        fields = [w] + ([str(rng.randrange(1, 90))] if with_freq else [])
        for _ in range(rng.randrange(1, 4)):
            cs = list(w)
            cs[rng.randrange(len(cs))] = rng.choice("aeiouy")
            if rng.random() < 0.4:
                cs.insert(rng.randrange(len(cs) + 1), rng.choice("hst"))
            fields += ["".join(cs), str(rng.choice([1.0, 0.9, 0.75, 0.5]))] + ([str(rng.randrange(1, 50))] if with_freq else [])
        lines.append("\t".join(fields))
    lines.append(lines[0])  # a repeated line: duplicate links
    open(path, "w", encoding="utf-8").write("\n".join(lines) + "\n")


def test0801_expand_variants():
    for mk in ("twin", "c"):
        if mk == "twin":
            m = T.VariantModel(T.TEST_ALPHABET)
            vid = m.add_to_vocabulary("afgescheid")
            m.add_variant(vid, "afghescheydt", 1.0, None, transparent=True)
            m.build()
            r = m.find_variants("afgheschaydt", T.test_searchparams())
            got = [(m.decoder[x.vocab_id].text, m.decoder[x.via].text) for x in r]
        else:
            m = O.OracleModel(alphabet_text=TEST_ALPHABET_TSV)
            vid = m.add("afgescheid")
            m.add_variant(vid, "afghescheydt", 1.0, None, True)
            m.build()
            r = m.find_variants_via("afgheschaydt", O.make_params(("abs", 2), ("abs", 2), 10, 0.0, 0.0))
            got = [(m.text(v), m.text(via)) for v, _, _, via in r]
        assert got == [("afgescheid", "afghescheydt")]


def test_tutorial_variant_list(data_dir, tutorial_outputs, tmp_path):
    vl = tutorial_outputs["variant_list"]
    f = tmp_path / "example.variantlist.tsv"
    f.write_text(vl["file_content"], encoding="utf-8")
    alphabet = data_dir + "/simple.alphabet.tsv"
    tm = T.VariantModel(T.read_alphabet(alphabet))
    tm.read_variants(str(f), transparent=vl["transparent"])
    tm.build()
    om = O.OracleModel(alphabet_path=alphabet)
    om.read_variants(str(f), vl["transparent"])
    om.build()
    assert len(tm.index) == 3 and om.n_classes() == 3 and om.n_instances() == 3
    tp = T.SearchParameters(("abs", 2), ("abs", 2), 1)
    op = O.make_params(("abs", 2), ("abs", 2), 1)
    for case in vl["cases"]:
        r = tm.find_variants(case["input"], tp)
        assert [[tm.decoder[x.vocab_id].text, x.score(0.0), x.dist_score, x.freq_score, tm.decoder[x.via].text] for x in r] == case["results"]
        r = om.find_variants_via(case["input"], op)
        assert [[om.text(v), d, d, fq, om.text(via)] for v, d, fq, via in r] == case["results"]


def test_c_oracle_vs_twin_random_variant_lists(tmp_path):
    rng = random.Random(12)
    for trial in range(6):
        with_freq, transparent = trial % 2 == 1, trial % 3 != 0
        f = str(tmp_path / f"v{trial}.tsv")
        make_variant_file(f, rng, with_freq)
        tm = T.VariantModel(T.TEST_ALPHABET)
        om = O.OracleModel(alphabet_text=TEST_ALPHABET_TSV)
        for w in ("house", "mousse", "hoes"):
            tm.add_to_vocabulary(w)
            om.add(w)
        tm.read_variants(f, transparent)
        om.read_variants(f, transparent)
        tm.build()
        om.build()
        for n, thr, fw in ((10, 0.0, 0.0), (2, 0.3, 0.0), (0, 0.0, 0.0), (3, 0.2, 0.7)):
            tp = T.SearchParameters(("abs", 3), ("abs", 3), n, thr, 0.0 if fw else 2.0, False, fw)
            op = O.make_params(("abs", 3), ("abs", 3), n, thr, 0.0 if fw else 2.0, False, fw)
            for q in ("house", "hause", "mose", "huose", "shuot", "ours", "hsoe", "xyz"):
                a = [(x.vocab_id, x.dist_score, x.freq_score, x.via) for x in tm.find_variants(q, tp)]
                assert a == om.find_variants_via(q, op), (trial, q, n, fw)


def test_product_host_model_variants(tmp_path):
    m = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=-1)
    vid = m.add_to_vocabulary("afgescheid")
    assert m.add_variant(vid, "afghescheydt", 1.0, None, A.VocabParams(vocabtype="INDEXED|TRANSPARENT")) is True
    assert m.add_variant(vid, "afgescheid", 1.0) is False  # variant == reference, src/lib.rs:479,511
    m.build()
    assert m.num_instances() == 2 and "afghescheydt" in m
    f = tmp_path / "vl.tsv"
    f.write_text("separate\t10\tseperate\t1.0\t3\tseprate\t0.9\t2\n", encoding="utf-8")
    m2 = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=-1)
    m2.read_variants(str(f), transparent=True)
    m2.build()
    assert [m2.vocab_text(i) for i in (3, 4, 5)] == ["separate", "seperate", "seprate"]
    from analiticcl_amd import _lib as L
    assert [L.lib().anx_model_vocab_frequency(m2.h, i) for i in (3, 4, 5)] == [10, 3, 2]
