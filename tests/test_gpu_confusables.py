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


# ---- the weighting on the device (conf.hip, default) against the host threads (ANX_CONFUSABLES=host): same source for the edit
# script and the matcher (confusables_core.hpp), so every row must come out identical -- ids, order and f64 scores --------------------
CONF10 = os.path.join(synth.GOLDEN_DATA, "confusables10.tsv")


def _nld_model(data_dir, early, extra=()):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "nld.aspell.lexicon"))
    g.read_confusablelist(CONF10)
    for script, w in extra:
        g.add_to_confusables(script, w)
    if early:
        g.set_confusables_before_pruning()
    g.build()
    return g


def _both_modes(g, qs, p, early=False, compact=True):
    import numpy as np
    out = {}
    for mode in ("device", "host"):
        A.set_switch("ANX_CONFUSABLES", "host" if mode == "host" else None)
        try:
            b = g.encode_batch(qs, p)
            b.run()
            out[mode] = b.fetch_arrays()
            if mode == "device" and compact:   # final on the device: the compact download and a second run work too
                coff, crows = b.fetch_compact()
                assert np.array_equal(coff, out[mode][0]) and np.array_equal(crows["dist_score"], out[mode][2])
                b.run()
                for x, y in zip(out[mode], b.fetch_arrays()):
                    assert np.array_equal(x, y)
            b.free()
        finally:
            A.set_switch("ANX_CONFUSABLES", None)
    # (early mode too: the host path puts the rows back into the reference's gather order before it weights and sorts them)
    for x, y in zip(out["device"], out["host"]):
        assert np.array_equal(x, y)
    return out["device"]


@pytest.mark.parametrize("early", [False, True])
@pytest.mark.parametrize("kw", [dict(max_anagram_distance=3, max_edit_distance=3, max_matches=10),
                                dict(max_anagram_distance=3, max_edit_distance=2, max_matches=3, freq_weight=0.5, cutoff_threshold=1.5),
                                dict(max_anagram_distance=3, max_edit_distance=2, max_matches=0, score_threshold=0.4)])
def test_device_weighting_equals_host(data_dir, early, kw):
    import numpy as np
    words = synth.load_lexicon_words(os.path.join(data_dir, "nld.aspell.lexicon"))
    # multi-character and non-ASCII options, a `$` tail and `^`: the screen defers them to the exact matcher
    g = _nld_model(data_dir, early, extra=(("-[ij]+[y]", 1.07), ("=[e]-[ë]+[e]", 1.03), ("^-[s]+[z]", 0.93), ("-[en]$", 0.9)))
    qs = synth.make_queries(words, 150_000, max_len=24, seed=41) + ["", "x" * 300, "ijsvrij", "zeeën", "naïve", "lopen"]
    off, vid, dist, freq = _both_modes(g, qs, A.SearchParameters(**kw), early=early)
    assert off[-1] > 100_000 and (dist != np.round(dist, 12)).any()


def test_rows_the_device_cannot_weight_fall_back_to_the_host(data_dir):
    """An input of more than 64 code points that still has candidates (a long compound in the lexicon) exceeds the fixed working
    memory of conf.hip: the run raises the flag and the whole batch is redone with the host-side weighting -- same rows."""
    words = synth.load_lexicon_words(os.path.join(data_dir, "nld.aspell.lexicon"))
    long_word = "aansprakelijkheidswaardevaststellingsveranderingen" * 2      # 98 code points, in the lexicon below
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "nld.aspell.lexicon"))
    g.add_to_vocabulary(long_word)
    g.read_confusablelist(CONF10)
    g.build()
    qs = synth.make_queries(words, 20_000, max_len=20, seed=43) + [long_word[:-1] + "m", long_word]
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    off, vid, dist, freq = _both_modes(g, qs, p, compact=False)   # (after the fallback the rows are rescored on the host)
    assert off[-1] - off[-3] >= 2      # the long inputs found their entry


def test_exports_work_with_confusables_on_the_device(data_dir):
    import numpy as np
    import torch
    from analiticcl_amd import shard
    words = synth.load_lexicon_words(os.path.join(data_dir, "nld.aspell.lexicon"))
    g = _nld_model(data_dir, False)
    qs = synth.make_queries(words, 30_000, max_len=20, seed=47)
    b = g.encode_batch(qs, A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10))
    b.run()
    off, vid, dist, freq = b.fetch_arrays()
    buf = torch.empty(shard.compact_capacity(len(qs), 16), dtype=torch.uint8, device="cuda:0")
    used = b.export_compact(buf.data_ptr(), buf.numel())
    torch.cuda.synchronize()
    dec = shard.decode_compact(buf[:used], len(qs))
    for i in (0, 17, 29_999):
        assert [(v, d) for v, d, _f in dec[i]] == [(int(vid[j]), float(dist[j])) for j in range(off[i], off[i + 1])]
    b.free()
