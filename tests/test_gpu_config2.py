"""BASELINE.json configs[2] as stated, all at once: nld.aspell lexicon (~230 k entries), 1 M synthetic queries of length <= 24,
max-edit-distance 3, WITH confusable weighting (the committed 10-pattern list tests/golden/data/confusables10.tsv), on one
MI355X.  Checked through size-independent properties and >= 1000 spot checks against the oracle with confusables ON
(oracle/confusable_oracle.py: C oracle up to the crop + sesdiff twin rescoring, re-rank, cutoff; the composition is pinned to
the full twin by tests/test_confusables_cpu.py).  Reference: src/lib.rs:972-1027, :1591-1595, :1656-1663, :1733-1756."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import confusable_oracle as CO
from oracle import cwrap as O

from fullsize_common import check_ranked, check_shards_equal_whole, checksum

N = 1_000_000
CONF = os.path.join(synth.GOLDEN_DATA, "confusables10.tsv")


@pytest.fixture(scope="module")
def setup(data_dir):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "nld.aspell.lexicon"))
    g.read_confusablelist(CONF)
    g.build()
    words = synth.load_lexicon_words(os.path.join(data_dir, "nld.aspell.lexicon"))
    qs = synth.make_queries(words, N, max_len=24, seed=synth.SEED + 2)
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=3, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
    b = g.encode_batch(qs, p)
    b.run()
    return g, qs, p, b, b.fetch_arrays(), b.stats()


def test_counts_and_idempotence(setup):
    g, qs, p, b, (off, vid, dist, freq), st = setup
    # n_results = rows the device ranked (cropped, cutoff not applied: it follows the host-side confusable rescoring)
    assert st["n_queries"] == N and off[-1] <= st["n_results"] and off[-1] > 0.9 * st["n_results"]
    assert 50 * N < st["n_pairs"] < 400 * N
    c1 = checksum(off, vid, dist, freq)
    b.run()
    assert b.stats()["n_pairs"] == st["n_pairs"]
    assert checksum(*b.fetch_arrays()) == c1


def test_ranked_and_bounded(setup):
    _g, _qs, _p, _b, (off, vid, dist, freq), _st = setup
    # a penalised row (weight < 1) may fall below the score threshold it passed before the rescoring
    check_ranked(off, dist, N, 11, 0.25, 2.0, score_floor_exact=False)
    assert dist.min() >= 0.25 * 0.9 * 0.95 - 1e-12 and dist.max() <= 1.1 * 1.1 * 1.1 + 1e-12
    assert np.all(freq == 1.0)


def test_shards_equal_whole(setup):
    g, qs, p, _b, arrays, _st = setup
    check_shards_equal_whole(g, qs, p, arrays, ((0, 40_000), (700_001, 745_000)))


def test_oracle_spot_check_with_confusables(setup, data_dir):
    g, qs, _p, _b, (off, vid, dist, freq), _st = setup
    o = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
    o.read_lexicon(os.path.join(data_dir, "nld.aspell.lexicon"))
    o.build()
    confs = CO.read_confusables(CONF)
    op = O.make_params(("abs", 3), ("abs", 3), 10, 0.25, 0.0)  # the cutoff follows the late rescoring
    rng = np.random.default_rng(11)
    fired = 0
    # 20 000 of the million queries (round 5: 1 200): the C oracle's OpenMP batch entry up to the crop, then the sesdiff twin's
    # weights, re-rank and cutoff per query
    idx = [int(i) for i in rng.choice(N, 20_000, replace=False)]
    c, ov, od, of, _tp, _tc = O.batch_rows(o, [qs[i] for i in idx], op, nthreads=16, stride=16)
    for n, i in enumerate(idx):
        exp = CO.late_rescore([(int(ov[n, j]), float(od[n, j]), float(of[n, j])) for j in range(c[n])], qs[i], confs, o.text, 0.0, 2.0)
        got = [(int(vid[j]), float(dist[j]), float(freq[j])) for j in range(off[i], off[i + 1])]
        assert [v for v, _d, _f in got] == [v for v, _d, _f in exp], qs[i]
        for (_v, d, f), (_v2, d2, f2) in zip(got, exp):
            assert abs(d - d2) <= 1e-6 and f == f2, qs[i]  # north_star: float composite score within 1e-6
        fired += any(CO.confusable_weight(confs, qs[i], o.text(v)) != 1.0 for v, _d, _f in exp)
    assert fired > 1000  # the patterns did fire on the sample


def test_filters_and_adjacency_off_equal_default_with_confusables(setup):
    """configs[2] at full size WITH its confusable patterns loaded (the device weights the ranked rows): the run with the scan's fused
    band-match filter off (ANX_SCAN_FUSE=0), with no band-match bound at all (ANX_PREFILTER=0: every length-compatible pair goes
    through damerau_levenshtein, /root/reference/src/distance.rs:101-179) and with the probe walk instead of the adjacency lists
    (ANX_SCAN_ADJ=0) must agree with the default path on scored pairs, survivors and the checksum of the weighted rows."""
    g, qs, p, b, arrays, st = setup
    ref = (st["n_pairs"], st["n_survivors"], checksum(*arrays))
    assert st["n_prefiltered_in_scan"] > 0.5 * st["n_pairs"] and st["n_conf_scripts"] > 0 and st["n_adj_tiles"] > 0.9 * st["n_scan_blocks"]
    try:
        for sw in ("ANX_SCAN_FUSE", "ANX_PREFILTER"):
            A.set_switch(sw, "0")
            b.run()
            s2 = b.stats()
            assert (s2["n_pairs"], s2["n_survivors"], checksum(*b.fetch_arrays())) == ref, sw
            A.set_switch(sw, None)
        A.set_switch("ANX_SCAN_ADJ", "0")
        b2 = g.encode_batch(qs, p)   # read when the tiles are built
        b2.run()
        s3 = b2.stats()
        got = (s3["n_pairs"], s3["n_survivors"], checksum(*b2.fetch_arrays()))
        b2.free()
        assert s3["n_adj_tiles"] == 0 and got == ref
    finally:
        for sw in ("ANX_SCAN_FUSE", "ANX_PREFILTER", "ANX_SCAN_ADJ"):
            A.set_switch(sw, None)
