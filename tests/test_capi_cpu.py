"""CPU-side checks of the product library (no compute calls: there is no GPU in the build container).

 * libanx.so loads and exports every symbol include/anx.h declares;
 * host logic (alphabet, normalisation, anagram values, vocabulary ids, index build) matches the reference's
   known answers and the tutorial's recorded build statistics;
 * the query path fails loudly (ANX_ENODEVICE) instead of falling back to a CPU path.
"""
import ctypes
import os
import re

import pytest

import analiticcl_amd as A
from analiticcl_amd import _lib as L
from oracle import twin as T

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_ALPHABET_TSV = "\n".join(f"{c}\t{c.upper()}" for c in "abcdefghijklmnopqrstuvwxyz") + "\n.\t,\n"


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "anx.h")).read()
    names = sorted(set(re.findall(r"\b(anx_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 30
    lib = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert L.lib().anx_abi_version() == L.ABI_VERSION == 3


def test_defaults_match_reference():
    p = L.Params()
    L.lib().anx_default_params(ctypes.byref(p))  # src/types.rs:170-192
    assert (p.max_anagram_distance.kind, p.max_anagram_distance.value) == (0, 3)
    assert (p.max_edit_distance.kind, p.max_edit_distance.value) == (0, 3)
    assert (p.max_matches, p.score_threshold, p.cutoff_threshold, p.freq_weight) == (20, 0.25, 2.0, 0.0)
    w = L.Weights()
    L.lib().anx_default_weights(ctypes.byref(w))  # src/types.rs:57-67
    assert (w.ld, w.lcs, w.prefix, w.suffix, w.casew) == (0.5, 0.125, 0.125, 0.125, 0.125)
    v = L.VocabParams()
    L.lib().anx_default_vocab_params(ctypes.byref(v))  # src/vocab.rs:121-131
    assert (v.text_column, v.freq_column, v.freq_handling, v.vocab_type) == (0, 1, 1, 1)


def test_hash_and_normalize_known_answers():  # tests/main.rs:38-91, 559-563
    m = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=-1)
    h = m.anahash
    assert (h("a"), h("b"), h("c"), h("ab"), h("ba"), h("abc")) == (2, 3, 5, 6, 6, 30)
    assert h("abcabcabc") == 30 ** 3 and h("") == 1
    assert h("abc") == h("ABC") == h("bAc") and h("a.b") == h("a,b")
    assert h("xyz" * 24) == (89 * 97 * 101) ** 24
    assert h("stressed") == h("desserts") and h("dormitory") == h("dirtyroom")
    assert m.normalize("a") == [0] and m.normalize("b") == [1]
    assert m.normalize("a?b") == [0, 28, 1]  # UNK norm code = alphabet.len()+1, src/anahash.rs:76
    assert h("?") == T.PRIMES[27]  # UNK prime index = alphabet.len(), src/anahash.rs:42
    assert L.lib().anx_model_alphabet_size(m.h) == 28


def test_vocabulary_and_index_small():  # tests/main.rs:816-855
    m = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=-1)
    lex = ["rites", "tiers", "tires", "tries", "tyres", "rides", "brides", "dire"]
    ids = [m.add_to_vocabulary(w) for w in lex]
    assert ids == list(range(3, 11))  # 0,1,2 reserved: src/vocab.rs:145-181
    assert m.add_to_vocabulary("rites") == 3  # duplicates reuse the id, src/lib.rs:910-948
    m.build()
    assert all(w in m for w in lex) and "unknown" not in m
    assert m.num_instances() == 8 and m.num_classes() == 5
    assert [m.vocab_text(i) for i in (0, 1, 2)] == ["<bos>", "<eos>", "<unk>"]


def test_index_matches_tutorial_and_twin(data_dir, tutorial_outputs):
    m = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=-1)
    m.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    m.build()
    b = tutorial_outputs["build"]
    assert m.num_instances() == b["instances"] and m.num_classes() == b["anagrams"]
    assert {c: m.bucket_size(c) for c in range(256) if m.bucket_size(c)} == {int(k): v for k, v in b["histogram"].items()}
    alphabet = T.read_alphabet(os.path.join(data_dir, "simple.alphabet.tsv"))
    for w in ["separate", "Fo's", "K", "``", "''x", "naïve", "Æsop", "it's 4 o'clock!", "\x1b", "œuvre"]:
        assert m.normalize(w) == T.normalize_to_alphabet(w, alphabet), w
        assert m.anahash(w) == T.anahash(w, alphabet), w
    assert "separate" in m and "seperate" not in m


def test_no_cpu_fallback(data_dir):
    """Without a HIP device the query path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=-1)
    m.add_to_vocabulary("huis")
    with pytest.raises(A.AnxError) as e:
        m.find_variants("huys", A.SearchParameters())
    assert e.value.code == L.ANX_ENOTBUILT
    m.build()
    with pytest.raises(A.AnxError) as e:
        m.find_variants("huys", A.SearchParameters())
    assert e.value.code == L.ANX_ENODEVICE
    with pytest.raises(A.AnxError) as e:
        m.to_device(0)
    assert e.value.code == L.ANX_ENODEVICE


def test_python_api_surface():
    """Same names as /root/reference/analiticcl.pyi for the query path."""
    for cls, names in ((A.VariantModel, ["build", "add_to_vocabulary", "read_vocabulary", "read_lexicon",
                                         "__contains__", "find_variants", "find_variants_par", "find_all_matches"]),
                       (A.SearchParameters, ["get_max_anagram_distance", "get_edit_distance", "get_max_matches",
                                             "get_score_threshold", "get_cutoff_threshold", "get_freq_weight", "to_dict"]),
                       (A.Weights, ["get_ld", "set_ld", "get_case", "to_dict"])):
        for n in names:
            assert hasattr(cls, n), (cls, n)
    with pytest.raises(ValueError):
        A.SearchParameters(bogus=1)
    p = A.SearchParameters(max_anagram_distance=(0.3, 4), max_edit_distance=0.2)._c()
    assert (p.max_anagram_distance.kind, p.max_anagram_distance.value) == (2, 4) and p.max_edit_distance.kind == 1


def test_index_image_roundtrip(data_dir, tmp_path):
    """anx_model_save_index / anx_model_load_index: the image of a built model reloads to the same vocabulary ids,
    index statistics and lookups without read_vocabulary / build; wrong alphabets and truncated files are refused."""
    alphabet = os.path.join(data_dir, "simple.alphabet.tsv")
    vl = tmp_path / "variants.tsv"
    vl.write_text("separate\tseperate\t1.0\tseprate\t0.9\n", encoding="utf-8")
    m = A.VariantModel(alphabet, A.Weights(), device=-1)
    m.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    m.read_variants(str(vl), transparent=True)
    m.add_to_vocabulary("the cat", 7, A.VocabParams(vocabtype="LM"))
    m.build()
    img = str(tmp_path / "eng.anxidx")
    m.save_index(img)
    m2 = A.VariantModel(alphabet, A.Weights(), device=-1)
    m2.load_index(img)
    lib = L.lib()
    assert lib.anx_model_vocab_size(m2.h) == lib.anx_model_vocab_size(m.h)
    assert m2.num_classes() == m.num_classes() > 108802 and m2.num_instances() == m.num_instances()
    for c in range(1, 30):
        assert lib.anx_model_bucket_size(m2.h, c) == lib.anx_model_bucket_size(m.h, c)
    for vid in (3, 4, 1000, 60000, lib.anx_model_vocab_size(m.h) - 1):
        assert m2.vocab_text(vid) == m.vocab_text(vid)
        assert lib.anx_model_vocab_frequency(m2.h, vid) == lib.anx_model_vocab_frequency(m.h, vid)
    assert "separate" in m2 and "seperate" in m2 and "zzzzzz" not in m2
    assert m2.lexicons == m.lexicons
    wrong = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=-1)
    with pytest.raises(A.AnxError):
        wrong.load_index(img)
    data = open(img, "rb").read()
    open(img, "wb").write(data[: len(data) // 2])
    with pytest.raises(A.AnxError):
        A.VariantModel(alphabet, A.Weights(), device=-1).load_index(img)


def test_contextrules_host_side(tmp_path):
    """add_contextrule / read_contextrules (src/lib.rs:570-765, src/search.rs:413-459) are host logic: patterns, tags,
    tag offsets and the error cases, on a model without a device."""
    amph, rept = tmp_path / "amphibians.tsv", tmp_path / "reptiles.tsv"
    amph.write_text("salamander\t3\nfrog\t3\ntoad\t2\n")
    rept.write_text("lizard\t3\nsnake\t3\n")
    m = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=-1)
    m.read_lexicon(str(amph))
    m.read_lexicon(str(rept))
    m.add_contextrule("frog; ?; ^", 1.1, ["a", "b"], ["0:1", "1:"])
    m.add_contextrule("@amphibians.tsv; !@reptiles.tsv", 0.9)
    m.add_contextrule("frog|toad; !(snake|lizard)", 1.2, ["a"])
    assert m.tags == ["a", "b"]
    assert L.lib().anx_model_num_tags(m.h) == 2 and L.lib().anx_model_tag_name(m.h, 5) is None
    for bad in (("newt", 1.0, [], []), ("@missing.tsv", 1.0, [], []), ("frog", 1.0, ["t"], ["x:1"]), ("frog", 1.0, ["t"], ["0:y"]),
                ("frog", 1.0, [""], []), ("frog; ", 1.0, [], [])):
        with pytest.raises(A.AnxError):
            m.add_contextrule(*bad)
    rules = tmp_path / "rules.tsv"
    rules.write_text("# comment\n\nfrog; toad\t1.5\tpair\n\t1.0\nsnake\t0.8\tx; y\t0:1; :\n")
    m.read_contextrules(str(rules))
    assert m.tags == ["a", "b", "t", "", "pair", "x", "y"]  # tags are interned before a rule is rejected (src/lib.rs:680-700)
    for text in ("frog\n", "frog\tabc\n", "frog\t1.0\tx; y\n", "frog\t1.0\tx\t0:1; 1:1\n"):
        bad = tmp_path / "bad.tsv"
        bad.write_text(text)
        with pytest.raises(A.AnxError):
            m.read_contextrules(str(bad))
    with pytest.raises(A.AnxError):
        m.read_contextrules(str(tmp_path / "does-not-exist.tsv"))


def test_output_formatters_host_side():
    """anx_format_query_output on hand-made rows (no device): Rust's f64 Display, TSV / JSON shapes (bin:21-187)."""
    m = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=-1)
    ids = [m.add_to_vocabulary(w, 3) for w in ("separate", 'qu"ote')]
    rows = (L.Result * 3)()
    for r, (vid, d, f) in zip(rows, ((ids[0], 0.734375, 1.0), (ids[1], 1.0, 0.5), (ids[0], 1e-7, 0.25))):
        r.vocab_id, r.dist_score, r.freq_score, r.via = vid, d, f, L.ANX_NO_VIA
    offs = (ctypes.c_size_t * 3)(0, 2, 3)
    inputs = (ctypes.c_char_p * 2)(b"seperate", b'x"y')
    def fmt(js, fw, first=1):
        buf, ln = ctypes.c_void_p(), ctypes.c_size_t(0)
        L.check(L.lib().anx_format_query_output(m.h, inputs, 2, rows, offs, fw, js, 0, first, ctypes.byref(buf), ctypes.byref(ln)))
        s = ctypes.string_at(buf, ln.value).decode()
        L.lib().anx_string_free(buf)
        return s
    assert fmt(0, 0.0) == 'seperate\tseparate\t0.734375\t\tqu"ote\t1\t\nx"y\tseparate\t0.0000001\t\n'
    assert fmt(0, 1.0).startswith("seperate\tseparate\t0.8671875\t\tqu\"ote\t0.75\t\n")
    js = fmt(1, 0.0, 1)
    assert js.startswith('    { "input": "seperate", "variants": [ \n        { "text": "separate", "score": 0.734375, "dist_score": 0.734375, "freq_score": 1 },\n')
    assert '{ "text": "qu\\"ote", "score": 1, "dist_score": 1, "freq_score": 0.5 }\n    ] }\n    ,{ "input": "x\\"y"' in js
    assert fmt(1, 0.0, 7).startswith("    ,{")
