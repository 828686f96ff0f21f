"""GPU parity tests: the HIP path (through the C ABI, libanx.so) against the CPU oracle and the golden vectors.

Bar: bit-exact integer distances (ld, lcs, prefix, suffix, samecase), identical ranked vocab-id lists,
scores within 1e-6 (in practice identical f64).
"""
import os
import random

import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import cwrap as O

TOL = 1e-6
TEST_ALPHABET_TSV = "\n".join(f"{c}\t{c.upper()}" for c in "abcdefghijklmnopqrstuvwxyz") + "\n.\t,\n"


def sp(th):
    """oracle-style threshold tuple -> python API value"""
    return th[1] if th[0] == "abs" else (float(th[1]) if th[0] == "ratio" else (float(th[1]), int(th[2])))


def params_pair(k=("abs", 3), d=("abs", 3), n=20, thr=0.25, cutoff=2.0, stop=False, fw=0.0):
    return (A.SearchParameters(max_anagram_distance=sp(k), max_edit_distance=sp(d), max_matches=n, score_threshold=thr,
                               cutoff_threshold=cutoff, stop_criterion=stop, freq_weight=fw),
            O.make_params(k, d, n, thr, cutoff, stop, fw))


CLI = dict(k=("abs", 3), d=("abs", 2), n=10, thr=0.25, cutoff=2.0)  # src/bin/analiticcl.rs:805-817


@pytest.fixture(scope="module")
def eng(data_dir):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    g.build()
    o = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
    o.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    o.build()
    return g, o


def compare_batch(g, o, queries, gp, op, check_pairs=True):
    b = g.encode_batch(queries, gp)
    b.run()
    res = b.fetch()
    stats = b.stats()
    # scored pairs per query as the PRODUCTION run's scan counts them (pairs failing the DL's length test are counted there
    # without being materialised); fetch_pairs below re-runs with every pair materialised
    counts = b.pair_counts()
    assert int(counts.sum()) == stats["n_pairs"]
    pairs_by_q = {}
    if check_pairs:
        for (q, vid, ld, lcs, pre, suf, same, _score) in b.fetch_pairs():
            pairs_by_q.setdefault(q, []).append((vid, ld, lcs if ld >= 0 else 0, pre if ld >= 0 else 0,
                                                 suf if ld >= 0 else 0, same if ld >= 0 else 1))
    total_pairs = 0
    for i, text in enumerate(queries):
        if text == "":
            assert res[i] == []
            continue
        ores, opairs, npairs, _ncls = o.find_variants(text, op, want_pairs=True, cap=1 << 17)
        total_pairs += npairs
        assert int(counts[i]) == npairs, (text, int(counts[i]), npairs)
        got = res[i]
        assert [v for v, _, _ in got] == [v for v, _, _ in ores], (text, got[:5], ores[:5])
        for (gv, gd, gf), (ov, od, of) in zip(got, ores):
            assert abs(gd - od) <= TOL and abs(gf - of) <= TOL, (text, gv, gd, od, gf, of)
            assert gd == od and gf == of, ("scores are expected to be identical f64", text, gd, od)
        if check_pairs:
            assert sorted(pairs_by_q.get(i, [])) == sorted(opairs), text
    assert stats["n_pairs"] == total_pairs
    b.free()
    return stats


def test_tutorial_outputs(eng, tutorial_outputs):
    g, _ = eng
    p = A.SearchParameters()
    cases = [(c["input"], c["results"]) for c in tutorial_outputs["find_variants"]]
    cases += [(m["input"], m["variants"]) for m in tutorial_outputs["find_all_matches"][0]["matches"]]
    cases += [(m["input"], m["variants"]) for m in tutorial_outputs["find_all_matches"][1]["matches"]]
    for text, exp in cases:
        got = [[r["text"], r["score"], r["dist_score"], r["freq_score"]] for r in g.find_variants(text, p)]
        assert got == exp, text
    r = g.find_variants("seperate", p)[0]
    assert r["lexicons"] and r["lexicons"][0].endswith("eng.aspell.lexicon")


def test_twin_vectors_all_paramsets(eng, twin_vectors):
    g, _ = eng
    for pname, v in twin_vectors["paramsets"].items():
        gp = A.SearchParameters(max_anagram_distance=sp(tuple(v["max_anagram_distance"])),
                                max_edit_distance=sp(tuple(v["max_edit_distance"])), max_matches=v["max_matches"],
                                score_threshold=v["score_threshold"], cutoff_threshold=v["cutoff_threshold"],
                                stop_criterion=v["stop_at_exact_match"], freq_weight=v["freq_weight"])
        cases = [c for c in twin_vectors["cases"] if c["params"] == pname]
        b = g.encode_batch([c["input"] for c in cases], gp)
        b.run()
        res = b.fetch()
        assert b.stats()["n_pairs"] == sum(c["n_pairs"] for c in cases), pname
        for c, r in zip(cases, res):
            got = [[g.vocab_text(v), v, ds, fs] for v, ds, fs in r]
            assert got == c["results"], (pname, c["input"])
        b.free()


def test_random_queries_vs_oracle_cli_defaults(eng, data_dir):
    g, o = eng
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    qs = synth.make_queries(words, 1500, max_len=16, seed=11)
    gp, op = params_pair(**CLI)
    st = compare_batch(g, o, qs, gp, op)
    assert st["n_queries"] == 1500 and st["n_pairs"] > 10000


@pytest.mark.parametrize("kw", [
    dict(k=("abs", 3), d=("abs", 3), n=20),                       # library defaults
    dict(k=("abs", 2), d=("abs", 2), n=10, thr=0.0, cutoff=0.0),  # src/test.rs:48-68
    dict(k=("ratio", 0.3), d=("ratiolimit", 0.25, 3), n=5, thr=0.5, cutoff=0.0),
    dict(k=("ratio", 1.0), d=("ratio", 0.5), n=3, thr=0.1, cutoff=1.0),
    dict(k=("abs", 3), d=("abs", 3), n=20, stop=True),
    dict(k=("abs", 2), d=("abs", 2), n=0, thr=0.0, cutoff=0.0),   # unlimited
    dict(k=("abs", 3), d=("abs", 3), n=1),
    dict(k=("abs", 4), d=("abs", 4), n=7, thr=0.3, cutoff=1.5),
    dict(k=("abs", 3), d=("abs", 2), n=10, fw=0.5),               # weighted ranking (all freqs 1 here)
    # weighted ranking with a tie at the crop boundary: the crop's tie branch compares dist_score with the weighted score of row
    # max_matches and cuts far before the cutoff point (found by tools/fuzz_confusables_device.py in round 3)
    dict(k=("abs", 2), d=("abs", 2), n=20, thr=0.5, cutoff=1.5, fw=0.5),
    dict(k=("abs", 3), d=("abs", 3), n=5, thr=0.3, cutoff=1.2, fw=0.25),
])
def test_parameter_sets_vs_oracle(eng, data_dir, kw):
    g, o = eng
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    qs = synth.make_queries(words, 250, max_len=20, seed=5)
    qs += ["Bono", "Carly", "rocket", "a", "I", "Fo", "K", "xyzzyq", "it's", "O'Neil", "étude", "naïve", "b2b", "``", "  ", "e.g.", "AAAA",
           "antidisestablishmentarianism", "pneumonoultramicroscopicsilicovolcanoconiosis", "ZZZZZZZZZZ"]
    gp, op = params_pair(**kw)
    compare_batch(g, o, qs, gp, op)


def test_edge_inputs(eng):
    g, o = eng
    gp, op = params_pair(**CLI)
    qs = ["", "a", "", "zzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzz", "the", "", "é", "\U0001F600x"]
    compare_batch(g, o, qs, gp, op)
    assert g.find_variants_ids([], gp) == []
    assert g.find_variants_ids([""], gp) == [[]]


def test_single_query_call_equals_batch(eng):
    g, _ = eng
    p = A.SearchParameters(max_edit_distance=2, max_matches=10)
    qs = ["seperate", "recieve", "teh", "acommodate", "wich"]
    batch = g.find_variants_ids(qs, p)
    for q, r in zip(qs, batch):
        assert g.find_variants_ids([q], p)[0] == r
    par = g.find_variants_par(qs, p)
    assert [x["input"] for x in par] == qs
    assert [[v["text"] for v in x["variants"]] for x in par] == [[g.vocab_text(v) for v, _, _ in r] for r in batch]


def test_export_topk_matches_fetch(eng):
    import numpy as np
    import torch
    g, _ = eng
    p = A.SearchParameters(max_edit_distance=2, max_matches=10)
    qs = ["seperate", "", "recieve", "teh", "xyzzyq", "acommodate"]
    b = g.encode_batch(qs, p)
    b.run()
    res = b.fetch()
    stride = 11
    buf = torch.zeros(len(qs) * stride * 16, dtype=torch.uint8, device="cuda:0")
    b.export_topk(buf.data_ptr(), stride)
    torch.cuda.synchronize()
    rec = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=np.dtype([("vocab_id", "<u4"), ("freq", "<f4"), ("dist", "<f8")]))
    rec = rec.reshape(len(qs), stride)
    for i, r in enumerate(res):
        if qs[i] == "":
            continue
        assert [int(x) for x in rec[i]["vocab_id"][:len(r)]] == [v for v, _, _ in r]
        assert all(int(x) == 0xFFFFFFFF for x in rec[i]["vocab_id"][len(r):])
        assert [float(x) for x in rec[i]["dist"][:len(r)]] == [d for _, d, _ in r]
    b.free()


def test_export_compact_matches_fetch(eng):
    """anx_batch_export_compact: offsets + unpadded records in a device buffer == the fetched rows; through the
    CompactGather helper (world 1) as bench.py uses it; too small a buffer is refused with the size needed."""
    import torch
    from analiticcl_amd import shard, synth
    g, _ = eng
    words = synth.load_lexicon_words(os.path.join(synth.GOLDEN_DATA, "eng_aspell.lexicon.gz"))
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
    qs = ["seperate", "", "recieve", "teh", "xyzzyq", "acommodate", ""] + synth.make_queries(words, 3000, max_len=16, seed=77)
    b = g.encode_batch(qs, p)
    b.run()
    res = b.fetch()
    gather = shard.CompactGather(shard.compact_capacity(len(qs), 11), "cuda:0", 0, 1)
    for step in range(3):
        buf = gather.acquire(step & 1)
        used = b.export_compact(buf.data_ptr(), buf.numel())
        gather.submit(step & 1, used)
    gather.flush()
    torch.cuda.synchronize()
    assert used == shard.compact_offsets_bytes(len(qs)) + 16 * sum(len(r) for r in res)
    for slot in (0, 1):
        got = shard.decode_compact(gather.result(slot)[0], len(qs))
        assert [[(v, d) for v, d, _ in r] for r in got] == [[(v, d) for v, d, _ in r] for r in res]
        assert all(abs(a[2] - c[2]) < 1e-6 for r, e in zip(got, res) for a, c in zip(r, e))
    small = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(A.AnxError):
        b.export_compact(small.data_ptr(), small.numel())
    b.free()
    e = g.encode_batch(["", ""], p)
    e.run()
    buf = torch.ones(64, dtype=torch.uint8, device="cuda:0")
    assert e.export_compact(buf.data_ptr(), 64) == 16 and shard.decode_compact(buf, 2) == [[], []]
    e.free()


def test_small_model_reference_tests():  # tests/main.rs:858-911
    lex = ["rites", "tiers", "tires", "tries", "tyres", "rides", "brides", "dire"]
    g = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=0)
    o = O.OracleModel(alphabet_text=TEST_ALPHABET_TSV)
    for w in lex:
        g.add_to_vocabulary(w)
        o.add(w)
    g.build()
    o.build()
    gp, op = params_pair(("abs", 2), ("abs", 2), 10, 0.0, 0.0)
    compare_batch(g, o, ["rite", "rites", "tyre", "bride", "x", "dir"], gp, op)
    g2 = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=0)
    for w in ("huis", "huls"):
        g2.add_to_vocabulary(w)
    g2.build()
    r = g2.find_variants("huys", gp)
    assert [x["text"] for x in r] == ["huis", "huls"]  # tied: order is the deterministic enumeration order
    assert r[0]["dist_score"] == r[1]["dist_score"] and r[0]["freq_score"] == r[1]["freq_score"]


def test_frequency_lexicon_and_weights(tmp_path):
    """have_freq lexicon (frequency column), freq_weight > 0, zero weights (components not computed)."""
    rng = random.Random(3)
    base = ["house", "houses", "horse", "hose", "mouse", "moose", "louse", "hours", "hour", "our", "ours", "use",
            "used", "user", "muse", "fuse", "ruse", "rouse", "douse", "nous", "thou", "shout", "south", "mouth"]
    lines = [f"{w}\t{rng.randrange(0, 500)}" for w in base] + ["hause", "Hause\t7"]
    lexfile = tmp_path / "freq.tsv"
    lexfile.write_text("\n".join(lines) + "\n", encoding="utf-8")
    qs = ["house", "hous", "huose", "mouse", "Hose", "ouse", "xouse", "thuo"]
    for weights in (A.Weights(), A.Weights(lcs=0.0, prefix=0.0), A.Weights(case=0.0, suffix=0.0, ld=1.0)):
        g = A.VariantModel("", weights, alphabet_text=TEST_ALPHABET_TSV, device=0)
        g.read_lexicon(str(lexfile))
        g.build()
        o = O.OracleModel(alphabet_text=TEST_ALPHABET_TSV)
        o.set_weights(weights.ld, weights.lcs, weights.prefix, weights.suffix, weights.case)
        o.read_lexicon(str(lexfile))
        o.build()
        for kw in (dict(fw=0.0), dict(fw=0.5), dict(fw=2.0, n=3), dict(fw=1.0, n=0, thr=0.0, cutoff=0.0),
                   dict(fw=0.25, n=2, cutoff=1.2)):
            gp, op = params_pair(("abs", 3), ("abs", 3), kw.get("n", 20), kw.get("thr", 0.25), kw.get("cutoff", 2.0),
                                 False, kw["fw"])
            compare_batch(g, o, qs, gp, op)


def test_nld_lexicon_edit_distance_3(data_dir):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "nld.aspell.lexicon"))
    g.build()
    o = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
    o.read_lexicon(os.path.join(data_dir, "nld.aspell.lexicon"))
    o.build()
    words = synth.load_lexicon_words(os.path.join(data_dir, "nld.aspell.lexicon"))
    qs = synth.make_queries(words, 400, max_len=24, seed=23)
    gp, op = params_pair(("abs", 3), ("abs", 3), 10, 0.25, 2.0)
    compare_batch(g, o, qs, gp, op)


# ---- variant lists / transparent entries ------------------------------------
def test_variant_list_reference_vectors(data_dir, tutorial_outputs, tmp_path):
    """tests/main.rs:1484-1510 (test0801) and tutorial.ipynb cells 27-32 through the HIP path."""
    g = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=0)
    vid = g.add_to_vocabulary("afgescheid")
    g.add_variant(vid, "afghescheydt", 1.0, None, A.VocabParams(vocabtype="INDEXED|TRANSPARENT"))
    g.build()
    p = A.SearchParameters(max_anagram_distance=2, max_edit_distance=2, max_matches=10, score_threshold=0.0,
                           cutoff_threshold=0.0)
    r = g.find_variants("afgheschaydt", p)
    assert [(x["text"], x["via"]) for x in r] == [("afgescheid", "afghescheydt")]
    vl = tutorial_outputs["variant_list"]
    f = tmp_path / "example.variantlist.tsv"
    f.write_text(vl["file_content"], encoding="utf-8")
    g2 = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g2.read_variants(str(f), transparent=True)
    g2.build()
    p2 = A.SearchParameters(**vl["params"])
    for case in vl["cases"]:
        got = [[x["text"], x["score"], x["dist_score"], x["freq_score"], x.get("via")] for x in g2.find_variants(case["input"], p2)]
        assert got == case["results"]
        assert g2.find_variants(case["input"], p2)[0]["lexicons"] == [str(f)]


def test_variant_lists_vs_oracle(tmp_path):
    from tests.test_variants_cpu import make_variant_file
    rng = random.Random(99)
    lexfile = tmp_path / "freq.tsv"
    lexfile.write_text("\n".join(f"{w}\t{rng.randrange(0, 60)}" for w in
                                 ["house", "mousse", "hoes", "horse", "hours", "shout", "south", "use"]) + "\n")
    queries = ["house", "hause", "mose", "huose", "shuot", "ours", "hsoe", "xyz", "mouse", "hoose", "rous", "us"]
    for trial in range(8):
        with_freq, transparent, with_lex = trial % 2 == 1, trial % 3 != 0, trial >= 4
        f = str(tmp_path / f"v{trial}.tsv")
        make_variant_file(f, rng, with_freq)
        g = A.VariantModel("", alphabet_text=TEST_ALPHABET_TSV, device=0)
        o = O.OracleModel(alphabet_text=TEST_ALPHABET_TSV)
        if with_lex:  # have_freq = true (src/lib.rs:544-547)
            g.read_lexicon(str(lexfile))
            o.read_lexicon(str(lexfile))
        g.read_variants(f, transparent)
        o.read_variants(f, transparent)
        g.build()
        o.build()
        for n, thr, cutoff, fw in ((10, 0.0, 0.0, 0.0), (2, 0.3, 2.0, 0.0), (0, 0.0, 0.0, 0.0), (3, 0.2, 0.0, 0.7), (1, 0.25, 2.0, 0.0)):
            gp, op = params_pair(("abs", 3), ("abs", 3), n, thr, cutoff, False, fw)
            got = g.find_variants_ids(queries, gp, with_via=True)
            for q, r in zip(queries, got):
                assert r == o.find_variants_via(q, op), (trial, q, n, fw)


def test_large_synthetic_lexicon_long_strings(data_dir, tmp_path):
    """Towards BASELINE.json configs[3]: eng.aspell + Markov-chain words (400 k entries, len 4-32), queries len 4-32:
    more classes per window, two-uint4 rows, the prefilter's pass-through for strings > 16 symbols."""
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    lex = synth.make_lexicon(words, 400_000, seed=7)
    f = tmp_path / "synth.lexicon"
    f.write_text("\n".join(lex) + "\n", encoding="utf-8")
    alphabet = os.path.join(data_dir, "simple.alphabet.tsv")
    g = A.VariantModel(alphabet, A.Weights(), device=0)
    g.read_lexicon(str(f))
    g.build()
    o = O.OracleModel(alphabet_path=alphabet)
    o.read_lexicon(str(f))
    o.build()
    assert g.num_classes() == o.n_classes() and g.num_instances() == o.n_instances()
    qs = synth.make_queries(lex, 300, max_len=32, min_len=4, seed=3)
    gp, op = params_pair(("abs", 3), ("abs", 2), 10, 0.25, 2.0)
    compare_batch(g, o, qs, gp, op)
    gp, op = params_pair(("ratio", 0.2), ("ratio", 0.15), 5, 0.3, 0.0)
    compare_batch(g, o, qs[:120], gp, op)


def test_very_long_strings(data_dir, tmp_path):
    """Strings of 20-70 symbols (compounds of lexicon words): pairs beyond the 32-symbol register kernels take the
    general k_score_pairs through the slot list; mixed with short words so that every scoring kernel sees work."""
    rng = random.Random(17)
    words = [w for w in synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon")) if w.isascii() and w.isalpha()][::29]
    lex = sorted({"".join(rng.choice(words) for _ in range(rng.randrange(2, 7))) for _ in range(4000)} | set(words[:1500]))
    f = tmp_path / "long.lexicon"
    f.write_text("\n".join(lex) + "\n", encoding="utf-8")
    alphabet = os.path.join(data_dir, "simple.alphabet.tsv")
    g = A.VariantModel(alphabet, A.Weights(), device=0)
    g.read_lexicon(str(f))
    g.build()
    o = O.OracleModel(alphabet_path=alphabet)
    o.read_lexicon(str(f))
    o.build()
    qs = synth.make_queries(lex, 400, max_len=80, min_len=1, seed=23)
    assert max(len(q) for q in qs) > 40
    for kw in (dict(k=("abs", 3), d=("abs", 3), n=10), dict(k=("abs", 4), d=("abs", 5), n=5, thr=0.2, cutoff=1.5),
               dict(k=("ratio", 0.1), d=("ratio", 0.1), n=20, thr=0.0, cutoff=0.0)):
        gp, op = params_pair(**kw)
        compare_batch(g, o, qs, gp, op)


def test_index_image_gives_identical_results(eng, data_dir, tmp_path):
    """A model loaded from anx_model_save_index's image answers exactly like the one that was built."""
    g, _ = eng
    img = str(tmp_path / "eng.anxidx")
    g.save_index(img)
    g2 = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g2.load_index(img)
    qs = synth.make_queries(synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon")), 500, max_len=20, seed=77)
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    assert g2.find_variants_ids(qs, p) == g.find_variants_ids(qs, p)
    assert g2.find_variants("seperate", p) == g.find_variants("seperate", p)


def test_oversized_call_is_split_into_device_batches(eng, data_dir, monkeypatch):
    """anx_find_variants_batch runs calls above its per-batch limit as consecutive device batches (limit lowered here)."""
    g, _ = eng
    qs = synth.make_queries(synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon")), 350, max_len=16, seed=8) + ["", "zzzzqq"]
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)

    def via_char_pp():  # the char** entry point (the Python Batch class uses the packed one)
        import ctypes as C
        from analiticcl_amd import _lib as L
        arr = (C.c_char_p * len(qs))(*[q.encode() for q in qs])
        cp = p._c()
        rows, offs = C.POINTER(L.Result)(), C.POINTER(C.c_size_t)()
        L.check(L.lib().anx_find_variants_batch(g.h, arr, len(qs), C.byref(cp), C.byref(rows), C.byref(offs)))
        out = [[(rows[j].vocab_id, rows[j].dist_score, rows[j].freq_score) for j in range(offs[i], offs[i + 1])] for i in range(len(qs))]
        L.lib().anx_results_free(rows, offs)
        return out
    whole = via_char_pp()
    A.set_switch("ANX_MAX_BATCH", 100)
    try:
        assert via_char_pp() == whole
    finally:
        A.set_switch("ANX_MAX_BATCH", None)


@pytest.mark.parametrize("nclasses", [40, 70, 110, 150])
def test_large_alphabets_take_the_count_vector_kernel(nclasses, tmp_path):
    """Alphabets with more than 31 classes cannot use the 32-bit thermometer planes: every tile runs k_scan_sad<NP>
    (NP = 16, 24, 32, 42 count-vector words).  Synthetic words over a nclasses-letter alphabet vs the oracle."""
    rng = random.Random(nclasses)
    letters = ([chr(c) for c in range(0x3B1, 0x3B1 + 25)] + [chr(c) for c in range(0x430, 0x430 + 32)] +
               [chr(c) for c in range(ord("a"), ord("z") + 1)] + [chr(c) for c in range(0x5D0, 0x5D0 + 27)] +
               [chr(c) for c in range(0x561, 0x561 + 38)] + [chr(c) for c in range(0x10D0, 0x10D0 + 33)])
    letters = letters[:nclasses]
    assert len(letters) == nclasses
    tsv = "\n".join(letters) + "\n"
    words = sorted({"".join(rng.choice(letters[: max(8, nclasses // 3)] if rng.random() < 0.5 else letters)
                            for _ in range(rng.randrange(2, 12))) for _ in range(6000)})
    g = A.VariantModel("", alphabet_text=tsv, device=0)
    o = O.OracleModel(alphabet_text=tsv)
    for w in words:
        g.add_to_vocabulary(w)
        o.add(w)
    g.build()
    o.build()
    qs = []
    for _ in range(300):
        cs = list(rng.choice(words))
        for _ in range(rng.randrange(0, 3)):
            op = rng.randrange(3)
            if op == 0 and len(cs) > 1:
                del cs[rng.randrange(len(cs))]
            elif op == 1:
                cs.insert(rng.randrange(len(cs) + 1), rng.choice(letters))
            else:
                cs[rng.randrange(len(cs))] = rng.choice(letters)
        qs.append("".join(cs))
    gp, op = params_pair(("abs", 3), ("abs", 2), 10, 0.2, 2.0)
    st = compare_batch(g, o, qs, gp, op)
    assert st["n_tests_kind"][0] > 0 and sum(st["n_tests_kind"][1:]) == 0


def test_concurrent_calls_from_host_threads(eng):
    """find_variants takes &self in the reference and is called from a rayon pool (src/bin/analiticcl.rs:445-448); the
    C entry point must therefore be callable from several host threads on one model at once (ctypes drops the GIL
    during the call).  Every thread's results equal the sequential ones."""
    import threading
    from analiticcl_amd import synth
    g, _ = eng
    words = synth.load_lexicon_words(os.path.join(synth.GOLDEN_DATA, "eng_aspell.lexicon.gz"))
    ps = [A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0),
          A.SearchParameters(max_anagram_distance=2, max_edit_distance=2, max_matches=3, score_threshold=0.1, cutoff_threshold=0.0),
          A.SearchParameters(max_anagram_distance=3, max_edit_distance=3, max_matches=20, stop_criterion=True)]
    jobs = [(synth.make_queries(words, 4000 + 500 * i, max_len=20, seed=100 + i), ps[i % 3]) for i in range(6)]
    expected = [g.find_variants_ids(q, p) for q, p in jobs]
    got = [None] * len(jobs)
    errors = []

    def work(i):
        try:
            for _ in range(3):
                got[i] = g.find_variants_ids(*jobs[i])
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    assert got == expected


def test_capacity_overflow_regrow_repeats_the_run(eng, monkeypatch):
    """A run is launched with capacity estimates (pair list, scoring grid, survivor list, slot lists, candidate rows) and read
    back once at its end; when an estimate did not hold, the buffers are regrown and the run repeated.  With the estimates
    divided by 64 (ANX_CAP_DIV) every one of them overflows: results must equal those of the normally sized run."""
    import numpy as np
    g, _o = eng
    words = synth.load_lexicon_words(os.path.join(synth.GOLDEN_DATA, "eng_aspell.lexicon.gz"))
    for max_len, kw in ((16, dict(max_anagram_distance=3, max_edit_distance=2, max_matches=10)),
                        (28, dict(max_anagram_distance=3, max_edit_distance=3, max_matches=10)),
                        (20, dict(max_anagram_distance=4, max_edit_distance=4, max_matches=5))):
        qs = synth.make_queries(words, 60000, max_len=max_len, seed=77)
        p = A.SearchParameters(**kw)
        b = g.encode_batch(qs, p)
        b.run()
        ref, st = b.fetch_arrays(), b.stats()
        b.free()
        A.set_switch("ANX_CAP_DIV", 64)
        try:
            b = g.encode_batch(qs, p)
            b.run()
            got, st2 = b.fetch_arrays(), b.stats()
            b.run()   # second run of the same batch: sized from the first
            got3 = b.fetch_arrays()
            b.free()
        finally:
            A.set_switch("ANX_CAP_DIV", None)
        for x, y, z in zip(ref, got, got3):
            assert np.array_equal(x, y) and np.array_equal(x, z)
        for k in ("n_pairs", "n_results", "n_survivors", "n_selected", "n_class_tests"):
            assert st[k] == st2[k], k


def test_async_runs_on_two_streams_equal_synchronous_runs(eng):
    """anx_batch_run_async / anx_batch_wait: two batches in flight on two HIP streams (the tail of one run under the scan of the
    other), relaunching a batch without waiting for its previous run, and fetching only after wait()."""
    import numpy as np
    import torch
    g, _o = eng
    words = synth.load_lexicon_words(os.path.join(synth.GOLDEN_DATA, "eng_aspell.lexicon.gz"))
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    qa, qb = synth.make_queries(words, 50000, max_len=16, seed=5), synth.make_queries(words, 70000, max_len=24, seed=6)
    ref = []
    for qs in (qa, qb):
        b = g.encode_batch(qs, p)
        b.run()
        ref.append(b.fetch_arrays())
        b.free()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ba, bb = g.encode_batch(qa, p), g.encode_batch(qb, p)
    for _ in range(3):
        ba.run_async(s1.cuda_stream)
        bb.run_async(s2.cuda_stream)
    with pytest.raises(A.AnxError):
        ba.fetch_arrays()                     # not waited for yet
    ba.run_async(s1.cuda_stream)              # relaunch without wait: the library waits for the previous run itself
    ba.wait()
    bb.wait()
    for b, r in ((ba, ref[0]), (bb, ref[1])):
        got = b.fetch_arrays()
        for x, y in zip(got, r):
            assert np.array_equal(x, y)
        b.free()


def test_host_threads_share_one_model(eng):
    """Four host threads, each encoding (packed entry point: device-side offsets + encoder on the null stream), running (own HIP
    stream) and fetching its own batches of different shapes on ONE model, several times over: every result equals the one the
    same batch gives alone.  (The device pool, the pinned result cache and the per-thread error state are what is shared.)"""
    import threading

    import numpy as np
    import torch
    g, _o = eng
    words = synth.load_lexicon_words(os.path.join(synth.GOLDEN_DATA, "eng_aspell.lexicon.gz"))
    shapes = [(30000, 16, 3, 2, 11), (45000, 24, 3, 3, 12), (20000, 12, 2, 1, 13), (60000, 16, 4, 2, 14)]
    jobs = []
    for n, maxlen, k, d, seed in shapes:
        qs = synth.make_queries(words, n, max_len=maxlen, seed=seed)
        p = A.SearchParameters(max_anagram_distance=k, max_edit_distance=d, max_matches=10)
        b = g.encode_batch(qs, p)
        b.run()
        ref = b.fetch_arrays()
        b.free()
        jobs.append((("\0".join(qs) + "\0").encode(), n, p, ref))
    errors = []

    def worker(job):
        packed, n, p, ref = job
        try:
            st = torch.cuda.Stream()
            for _ in range(4):
                b = g.encode_packed(packed, n, p)
                b.run(st.cuda_stream)
                got = b.fetch_arrays()
                for x, y in zip(got, ref):
                    assert np.array_equal(x, y)
                del got
                b.free()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_fetch_compact_equals_fetch(eng, data_dir, tmp_path):
    """anx_batch_fetch_compact: the ranked rows as 16-byte records + u32 offsets (half the PCIe bytes) carry the same ids, the
    same f64 dist scores and the f32 rounding of the freq scores; refused where a record cannot hold the row (variant lists:
    `via`; confusables weighted on the host)."""
    import numpy as np
    g, _o = eng
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    qs = ["", "seperate"] + synth.make_queries(words, 50000, max_len=20, seed=91) + ["x" * 300, ""]
    for kw in (dict(max_anagram_distance=3, max_edit_distance=2, max_matches=10), dict(max_anagram_distance=3, max_edit_distance=3, max_matches=0, freq_weight=0.3)):
        b = g.encode_batch(qs, A.SearchParameters(**kw))
        b.run()
        off, vid, dist, freq = b.fetch_arrays()
        coff, rows = b.fetch_compact()
        assert coff.dtype == np.uint32 and np.array_equal(coff, off)
        assert np.array_equal(rows["vocab_id"], vid) and np.array_equal(rows["dist_score"], dist)
        assert np.array_equal(rows["freq_score"], freq.astype(np.float32))
        b.free()
        del rows, coff
    b = g.encode_batch([], A.SearchParameters())
    b.run()
    coff, rows = b.fetch_compact()
    assert list(coff) == [0] and rows.size == 0
    b.free()
    m = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    m.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    m.add_to_confusables("-[y]+[i]", 1.1)
    m.build()
    A.set_switch("ANX_CONFUSABLES", "host")   # weighted on the host threads: the rows are not final on the device
    try:
        b = m.encode_batch(qs[:100], A.SearchParameters())
        b.run()
        with pytest.raises(A.AnxError, match="confusables"):
            b.fetch_compact()
        b.free()
    finally:
        A.set_switch("ANX_CONFUSABLES", None)
    b = m.encode_batch(qs[:100], A.SearchParameters())   # default: weighted on the device (conf.hip), compact rows are final
    b.run()
    off, vid, dist, _freq = b.fetch_arrays()
    coff, rows = b.fetch_compact()
    assert np.array_equal(coff, off) and np.array_equal(rows["vocab_id"], vid) and np.array_equal(rows["dist_score"], dist)
    b.free()


def test_pipeline_equals_staged_calls(eng, data_dir):
    """anx_pipeline: several packed batches in flight (encode / run / fetch on three library threads) return, in submission order, the
    rows of the synchronous staged calls; a job that fails reports its own error and the following jobs are unaffected."""
    import numpy as np
    eng = eng[0]
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    sets = [synth.make_queries(words, n, max_len=16, seed=100 + i) for i, n in enumerate((30_000, 1, 70_000, 500, 120_000, 30_000))]
    blobs = [b"".join(q.encode("utf-8") + b"\0" for q in qs) for qs in sets]
    want = []
    for qs, blob in zip(sets, blobs):
        b = eng.encode_packed(blob, len(qs), p)
        b.run()
        off, rows = b.fetch_compact()
        want.append((off.copy(), rows.copy()))
        b.free()
    pl = A.Pipeline(eng, depth=3)
    for rep in range(2):
        got = []
        for i, (qs, blob) in enumerate(zip(sets, blobs)):
            pl.submit(blob, len(qs), p)
            if pl.pending() == 3:
                got.append(pl.next())
        while pl.pending():
            got.append(pl.next())
        assert len(got) == len(sets)
        for (o, r), (wo, wr) in zip(got, want):
            assert np.array_equal(o, wo) and np.array_equal(r, wr)
    pl.submit(blobs[0], len(sets[0]) + 5, p)     # more strings announced than present: that job fails ...
    pl.submit(blobs[1], len(sets[1]), p)
    with pytest.raises(A.AnxError, match="fewer strings than announced"):
        pl.next()
    o, r = pl.next()                             # ... the next one does not
    assert np.array_equal(o, want[1][0]) and np.array_equal(r, want[1][1])
    for i in range(3):                           # a full pipeline refuses the next job instead of blocking its one caller thread
        pl.submit(blobs[i], len(sets[i]), p)
    with pytest.raises(A.AnxError, match="pipeline full"):
        pl.submit(blobs[3], len(sets[3]), p)
    assert pl.pending() == 3
    o, r = pl.next()
    assert np.array_equal(o, want[0][0]) and np.array_equal(r, want[0][1])
    pl.submit(blobs[3], len(sets[3]), p)         # jobs still in flight when the pipeline is freed
    pl.close()


@pytest.mark.parametrize("nclasses", [26, 61, 62])
def test_dl_by_diagonals_edge_cases(nclasses):
    """k_filter_score's Damerau-Levenshtein runs by diagonals on mismatch masks (kernels_score.hpp dl_diag), the masks from symbol planes
    for alphabets of <= 61 classes (26, 61) and from the byte rows beyond (62).  Words of 13..17 symbols (16 = the last length of the
    inline path, 17 = the first one of the 8-word kernel) over few letters, among them the alphabet's last ones (the highest symbol
    codes); queries made of them by transpositions with and without a gap, edits at the very start / end, repeated symbols; d = 1, 2, 3.
    Every scored pair's (ld, lcs, prefix, suffix) and the ranked lists against the oracle (src/distance.rs:101-231)."""
    rng = random.Random(900 + nclasses)
    letters = [chr(c) for c in range(ord("a"), ord("z") + 1)] + [chr(c) for c in range(0x3B1, 0x3B1 + 25)] + [chr(c) for c in range(0x430, 0x430 + 32)]
    letters = letters[:nclasses]
    assert len(letters) == nclasses
    tsv = "\n".join(letters) + "\n"
    few = letters[:5] + letters[-3:]
    words = set()
    while len(words) < 1200:
        words.add("".join(rng.choice(few) for _ in range(rng.choice([13, 14, 15, 16, 16, 16, 17]))))
    words = sorted(words)

    def mutate(w):
        w = list(w)
        for _ in range(rng.randint(1, 3)):
            op = rng.randrange(6)
            if op == 0 and len(w) > 2:      # adjacent transposition, often at an end
                i = rng.choice([0, len(w) - 2, rng.randrange(len(w) - 1)])
                w[i], w[i + 1] = w[i + 1], w[i]
            elif op == 1 and len(w) > 3:    # transposition around a deleted symbol
                i = rng.randrange(len(w) - 2)
                w[i], w[i + 2] = w[i + 2], w[i]
                del w[i + 1]
            elif op == 2 and len(w) > 2:    # transposition around an inserted symbol
                i = rng.randrange(len(w) - 1)
                w[i], w[i + 1] = w[i + 1], w[i]
                w.insert(i + 1, rng.choice(few))
            elif op == 3 and len(w) > 1:
                del w[rng.choice([0, len(w) - 1])]
            elif op == 4:
                w.insert(rng.choice([0, len(w)]), rng.choice(few))
            else:
                w[rng.randrange(len(w))] = rng.choice(few)
        return "".join(w)
    queries = [mutate(rng.choice(words)) for _ in range(500)] + words[:100]
    g = A.VariantModel("", alphabet_text=tsv, device=0)
    o = O.OracleModel(alphabet_text=tsv)
    for w in words:
        g.add_to_vocabulary(w)
        o.add(w)
    g.build()
    o.build()
    for d in (1, 2, 3):
        gp, op = params_pair(("abs", 3), ("abs", d), 50, 0.0, 0.0)
        st = compare_batch(g, o, queries, gp, op)
        assert st["n_survivors"] > 0

