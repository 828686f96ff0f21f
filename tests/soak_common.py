"""One round of the randomised soaks, shared by the tools (tools/fuzz_parity.py, tools/fuzz_search.py: open-ended runs on the GPU
box) and by tests/test_gpu_soak.py (a fixed-seed slice of them inside `pytest -m gpu`).  TEST INFRASTRUCTURE: the product is
compared with the C oracle / the Python twin."""
import os
import random

import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import cwrap as O

_state = {}


def _world():
    if not _state:
        d = synth.materialize_golden(f"/tmp/anxdata_soak_{os.getuid()}")
        _state["alpha"] = d["alphabet"]
        _state["words"] = [w for w in synth.load_lexicon_words(d["eng"])] + [w for w in synth.load_lexicon_words(d["nld"])][::3]
    return _state["alpha"], _state["words"]


def sp(th):
    return th[1] if th[0] == "abs" else (float(th[1]) if th[0] == "ratio" else (float(th[1]), int(th[2])))


def parity_round(seed, max_words=30000):
    """Random sub-lexicon (frequencies, variant lists), random weights and parameters, random queries: ranked ids, f64 scores and
    per-query scored-pair counts of the product against the C oracle."""
    ALPHA, allwords = _world()
    rng = random.Random(seed)
    nwords = min(max_words, rng.choice((300, 2000, 8000, 30000)))
    words = rng.sample(allwords, nwords)
    with_freq = rng.random() < 0.6
    w = [rng.choice((1.0, 0.5, 0.0)), rng.choice((1.0, 0.5, 0.0)), rng.choice((1.0, 0.0)), rng.choice((1.0, 0.0)), rng.choice((1.0, 0.2, 0.0))]
    if w[0] + w[1] + w[2] + w[3] + w[4] == 0:
        w[0] = 1.0
    g = A.VariantModel(ALPHA, A.Weights(ld=w[0], lcs=w[1], prefix=w[2], suffix=w[3], case=w[4]), device=0)
    o = O.OracleModel(alphabet_path=ALPHA)
    o.set_weights(*w)
    for x in words:
        f = rng.choice((1, 1, 2, 5, 40, 1000, 250000)) if with_freq else None
        if f is None:
            g.add_to_vocabulary(x)
            o.add(x)
        else:
            g.add_to_vocabulary(x, f)
            o.add(x, f)
    nvar = 0
    if rng.random() < 0.3:  # variant lists: weighted variants of some items (src/lib.rs:1677-1727)
        for x in rng.sample(words, min(200, nwords // 4)):
            ref = g.add_to_vocabulary(x, 3) if with_freq else g.add_to_vocabulary(x)
            oref = o.add(x, 3) if with_freq else o.add(x)
            assert ref == oref
            v = synth.make_queries([x], 1, max_len=40, seed=rng.randrange(1 << 30))[0]
            sc = rng.choice((1.0, 0.9, 0.5))
            tr = rng.random() < 0.3
            a = g.add_variant(ref, v, sc, None, A.VocabParams(vocabtype="INDEXED|TRANSPARENT") if tr else None)
            b = o.add_variant(oref, v, sc, None, tr)
            assert bool(a) == bool(b), (x, v)
            nvar += 1
    g.build()
    o.build()
    k = rng.choice((("abs", rng.randrange(1, 6)), ("ratio", rng.choice((0.2, 0.34, 0.5))), ("ratiolimit", 0.4, rng.randrange(1, 5))))
    dd = rng.choice((("abs", rng.randrange(1, 6)), ("ratio", rng.choice((0.2, 0.34, 0.5))), ("ratiolimit", 0.4, rng.randrange(1, 4))))
    n = rng.choice((0, 1, 2, 3, 10, 20))
    thr = rng.choice((0.0, 0.25, 0.5, 0.7))
    cutoff = rng.choice((0.0, 1.0, 1.5, 2.0, 3.0))
    stop = rng.random() < 0.25
    fw = rng.choice((0.0, 0.0, 0.5, 1.0))
    gp = A.SearchParameters(max_anagram_distance=sp(k), max_edit_distance=sp(dd), max_matches=n, score_threshold=thr,
                            cutoff_threshold=cutoff, stop_criterion=stop, freq_weight=fw)
    op = O.make_params(k, dd, n, thr, cutoff, stop, fw)
    nq = rng.choice((200, 600, 1500))
    qs = synth.make_queries(words, nq, max_len=rng.choice((12, 16, 24, 40)), seed=seed)
    qs += rng.sample(words, min(50, nwords))
    qs += ["".join(rng.choice("etaoins'. -éßAZ") for _ in range(rng.randrange(0, 30))) for _ in range(40)]
    qs += ["", "a", "e" * 70, "x" * 255, "y" * 256]
    b = g.encode_batch(qs, gp)
    b.run()
    res = b.fetch()
    counts = b.pair_counts()
    st = b.stats()
    b.free()
    total = 0
    for i, text in enumerate(qs):
        if text == "" or len(text) > 255:
            assert res[i] == [], (seed, text)
            continue
        ores, _opairs, npairs, _ = o.find_variants(text, op, want_pairs=True, cap=1 << 17)
        total += npairs
        assert int(counts[i]) == npairs, (seed, text, int(counts[i]), npairs)
        assert [(v, ds, fs) for v, ds, fs in res[i]] == [tuple(x) for x in ores], (seed, text, res[i][:4], ores[:4])
    assert st["n_pairs"] == total, (seed, st["n_pairs"], total)
    # the same configuration through the call at the reference's granularity (anx_find_variants_batch, char**): a random handful of
    # the queries -- the small path (small_path.hpp) answers when the model has no variant lists and the round no StopAtExactMatch, the
    # batch pipeline otherwise (and for the subset that keeps an input beyond 64 bytes): the rows must be the batch's either way
    sub = rng.sample(range(len(qs)), min(len(qs), rng.choice((1, 3, 40, 700))))
    if rng.random() < 0.7:
        sub = [i for i in sub if len(qs[i].encode("utf-8")) <= 64] or [1]
    got = small_call(g, [qs[i] for i in sub], gp)
    for i, r in zip(sub, got):
        exp = [] if (qs[i] == "" or len(qs[i]) > 255) else [(v, ds, fs) for v, ds, fs in res[i]]
        assert r == exp, (seed, "small call", qs[i], r[:3], exp[:3])
    return nwords, with_freq, nvar, len(qs), total, (k, dd, n, thr, cutoff, stop, fw)


def small_call(model, qs, params):
    """anx_find_variants_batch(char**) -> [[(vocab_id, dist, freq)]]"""
    import ctypes as C
    from analiticcl_amd import _lib as L
    lib = L.lib()
    arr = (C.c_char_p * len(qs))(*[q.encode("utf-8") for q in qs])
    rows = C.POINTER(L.Result)()
    offs = C.POINTER(C.c_size_t)()
    cp = params._c()
    L.check(lib.anx_find_variants_batch(model.h, arr, len(qs), C.byref(cp), C.byref(rows), C.byref(offs)))
    try:
        off = [offs[i] for i in range(len(qs) + 1)]
        return [[(rows[j].vocab_id, rows[j].dist_score, rows[j].freq_score) for j in range(off[i], off[i + 1])] for i in range(len(qs))]
    finally:
        lib.anx_results_free(rows, offs)




def search_round(seed, worlds):
    """Fresh random texts (short and long stretches, varying max_seq, with / without LM and context rules) through
    anx_find_all_matches_batch: every Match field against the Python twin."""
    import test_gpu_search as T
    rng = random.Random(seed)
    with_lm, with_rules = rng.random() < 0.6, rng.random() < 0.4
    if (with_lm, with_rules) not in worlds:
        worlds[(with_lm, with_rules)] = T.build_world(with_lm, with_rules)
    g, tw, words, phrases = worlds[(with_lm, with_rules)]
    texts = T.random_texts(words[:400] if with_lm else words, phrases, 60, seed)
    pool = synth.make_queries(words[:400], 400, max_len=14, seed=seed + 7)
    texts += [" ".join(pool[i:i + rng.randrange(10, 80)]) for i in range(0, 300, 80)]  # stretches without a hard boundary
    max_seq = rng.choice((1, 2, 5, 20, 40, 250))
    # every second round as concurrent parts of the call (search.cpp: large calls only, so the threshold comes down for the soak)
    parts = rng.choice((None, None, "2", "3", "5"))
    if parts:
        A.set_switch("ANX_SEARCH_PARTS_MIN", "1")
        A.set_switch("ANX_SEARCH_PARTS", parts)
    try:
        n_multi, n_tagged = T.compare_with_twin(g, tw, texts, max_seq)
    finally:
        if parts:
            A.set_switch("ANX_SEARCH_PARTS_MIN", None)
            A.set_switch("ANX_SEARCH_PARTS", None)
    return with_lm, with_rules, max_seq, len(texts), n_multi, n_tagged


def conf_round(seed):
    """Device-vs-host round of the confusable weighting: a random pattern set over a golden lexicon, random parameters, late or early
    mode; every ranked row (ids, order, f64 scores) of the device path (conf.hip) must equal the host path (ANX_CONFUSABLES=host);
    both compile confusables_core.hpp, the device through its kernels' own memory layout.  Returns the ranked rows compared."""
    import numpy as np
    rng = random.Random(seed)
    d = synth.materialize_golden(f"/tmp/anxdata_soak_{os.getuid()}")
    letters = "abcdefghijklmnopqrstuvwxyz"
    lex = rng.choice(["eng", "nld"])
    words = _state.setdefault("lexwords_" + lex, synth.load_lexicon_words(d[lex]))
    g = A.VariantModel(d["alphabet"], A.Weights(), device=0)
    g.read_lexicon(d[lex])
    pats = []
    for _ in range(rng.randrange(1, 14)):
        ops = []
        for _k in range(rng.randrange(1, 4)):
            op = rng.choice("-+=")
            opts = "|".join("".join(rng.choice(letters + "\u00eb\u00e9\u00ef") for _ in range(rng.choice([1, 1, 1, 2]))) for _ in range(rng.choice([1, 1, 2, 3])))
            ops.append(f"{op}[{opts}]")
        script = ("^" if rng.random() < 0.15 else "") + "".join(ops) + ("$" if rng.random() < 0.15 else "")
        pats.append((script, rng.choice([0.8, 0.9, 0.95, 1.05, 1.1, 1.2])))
        g.add_to_confusables(*pats[-1])
    early = rng.random() < 0.4
    if early:
        g.set_confusables_before_pruning()
    g.build()
    qs = synth.make_queries(words, rng.choice([100_000, 300_000]), max_len=rng.choice([12, 16, 24, 30]), seed=rng.randrange(1 << 30))
    p = A.SearchParameters(max_anagram_distance=rng.choice([2, 3]), max_edit_distance=rng.choice([1, 2, 3]), max_matches=rng.choice([0, 1, 3, 10, 20]),
                           score_threshold=rng.choice([0.0, 0.25, 0.5]), cutoff_threshold=rng.choice([0.0, 1.5, 2.0]), freq_weight=rng.choice([0.0, 0.0, 0.5]))
    out = {}
    try:
        for mode in ("device", "host"):
            A.set_switch("ANX_CONFUSABLES", "host" if mode == "host" else None)
            b = g.encode_batch(qs, p)
            b.run()
            out[mode] = b.fetch_arrays()
            b.free()
    finally:
        A.set_switch("ANX_CONFUSABLES", None)
    for x, y in zip(out["device"], out["host"]):
        assert np.array_equal(x, y), (seed, lex, "early" if early else "late", pats)
    return int(out["device"][0][-1])
