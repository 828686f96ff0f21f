"""BASELINE.json configs[4], one GPU's share: search mode over bulk running text -- 12.5 MB (100 MB / 8 GPUs) of synthetic
sentences of 5-25 perturbed eng.aspell words (analiticcl_amd/synth.py make_running_text), n-gram windowing up to order 3,
bigram LM rerank, the full find_all_matches path through anx_find_all_matches_batch (one device batch per n-gram order over
ALL texts).  Checked through size-independent properties (byte offsets, ordering, shard == whole) and a spot check of 40
sampled texts (320 sentences) against the oracle twin's search mode (segmentation / lattice / LM unchanged, per-segment
find_variants answered by the C oracle).  Reference: src/lib.rs:1790-1957, 2088-2495, 2580-2674; src/search.rs:190-336."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import cwrap as O
from oracle import twin as T

from search_common import twin_matches_parallel

MB = 12.5
LM = A.VocabParams(vocabtype="LM")


def _lm_entries(words):
    rng = random.Random(7)
    common = [w for w in words if w.isalpha()][::23][:5000]
    out = []
    for _ in range(20000):
        a, b = rng.choice(common), rng.choice(common)
        out.append((f"{a} {b}", rng.randrange(1, 20)))
    out += [(f"<bos> {w}", 5) for w in common[:500]]
    return common, out


@pytest.fixture(scope="module")
def setup(data_dir):
    lex = os.path.join(data_dir, "eng.aspell.lexicon")
    words = synth.load_lexicon_words(lex)
    common, lm = _lm_entries(words)
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(lex)
    for t, f in lm:
        g.add_to_vocabulary(t, f, LM)
    g.build()
    texts = synth.make_running_text(common, MB, seed=7)
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0,
                           max_ngram=3)
    return g, lex, lm, texts, p, g.find_all_matches_arrays(texts, p)


def test_bulk_offsets_and_structure(setup):
    _g, _lex, _lm, texts, _p, (off, ma, ra) = setup
    raw = [t.encode("utf-8") for t in texts]
    assert sum(len(r) for r in raw) >= MB * 1e6 and off.size == len(texts) + 1
    assert ma.size == off[-1] and ma.size > 1_000_000   # ~15 tokens per sentence, 8 sentences per text, some merged into n-grams
    tlen = np.repeat(np.array([len(r) for r in raw], dtype=np.uint64), np.diff(off))
    assert np.all(ma["begin"] < ma["end"]) and np.all(ma["end"] <= tlen)
    assert np.all((ma["n"] >= 1) & (ma["n"] <= 3))
    first = np.zeros(ma.size, dtype=bool)
    first[off[:-1][np.diff(off) > 0]] = True
    assert np.all(ma["begin"][1:][~first[1:]] >= ma["end"][:-1][~first[1:]])   # ordered, non-overlapping inside a text
    nv = (ma["ve"] - ma["vb"]).astype(np.int64)
    assert np.all(ma["ve"] >= ma["vb"]) and np.all(ma["ve"] <= ra.size) and nv.max() <= 11
    assert np.all((ma["selected"] >= -1) & (ma["selected"] < np.maximum(nv, 1)))
    assert np.all(ma["selected"][nv == 0] == -1) and np.all(ma["selected"][nv > 0] >= 0)
    assert np.all((ra["dist"] >= 0.25) & (ra["dist"] <= 1.0)) and (ma["n"] > 1).sum() > 0
    # the matched span never starts or ends with a separator of the generator (byte offsets land on token edges)
    rng = np.random.default_rng(1)
    tix = np.searchsorted(off, np.arange(ma.size), side="right") - 1
    for j in rng.choice(ma.size, 20000, replace=False):
        span = raw[tix[j]][int(ma["begin"][j]):int(ma["end"][j])]
        assert span and span[:1] not in b" \n.," and span[-1:] not in b" \n.,", (tix[j], span)
        assert span.count(b" ") + span.count(b"\n") == int(ma["n"][j]) - 1   # an n-gram spans n - 1 single-character separators


def test_shard_equals_whole(setup):
    """Texts are independent: a contiguous slice of them run as its own batch returns the matches of the whole run (the
    multi-GPU split of config 4)."""
    g, _lex, _lm, texts, p, (off, ma, ra) = setup
    lo, hi = 1000, 1600
    o2, m2, r2 = g.find_all_matches_arrays(texts[lo:hi], p)
    assert np.array_equal(o2, off[lo:hi + 1] - off[lo])
    whole = ma[off[lo]:off[hi]]
    for k in ("begin", "end", "n", "selected"):
        assert np.array_equal(m2[k], whole[k]), k
    assert np.array_equal(m2["ve"] - m2["vb"], whole["ve"] - whole["vb"])
    rw = ra[int(whole["vb"][0]):int(whole["ve"][-1])]
    assert np.array_equal(r2, rw)


def test_twin_spot_check(setup, data_dir):
    g, lex, lm, texts, p, (off, ma, ra) = setup
    # 400 sampled texts (3 200 sentences; round 5: 40 texts) through the oracle twin's search mode, spread over 16 child processes
    idx = random.Random(5).sample(range(len(texts)), 400)
    exp_all, _secs = twin_matches_parallel(os.path.join(data_dir, "simple.alphabet.tsv"), lex, lm, [texts[i] for i in idx], workers=16)
    n_multi = n_sent = 0
    for i, exp in zip(idx, exp_all):
        got = ma[off[i]:off[i + 1]]
        raw = texts[i].encode("utf-8")
        n_sent += 8
        assert [(raw[int(m["begin"]):int(m["end"])].decode(), int(m["begin"]), int(m["end"])) for m in got] == \
            [(e[0], e[1], e[2]) for e in exp], texts[i]
        for m, e in zip(got, exp):
            ev = e[5]
            rows = ra[int(m["vb"]):int(m["ve"])]
            assert [int(v) for v in rows["vocab_id"]] == [v[0] for v in ev], (texts[i], e[0])
            for r, w in zip(rows, ev):
                assert abs(float(r["dist"]) - w[1]) <= 1e-6 and abs(float(r["freq"]) - w[2]) <= 1e-6
            if ev:
                assert int(m["selected"]) == e[4], (texts[i], e[0])
            n_multi += e[3] > 1
    assert n_sent >= 3000 and n_multi > 0
