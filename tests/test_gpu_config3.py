"""BASELINE.json configs[3], one GPU's share: merged 1 M-entry synthetic lexicon (eng.aspell + nld.aspell + seeded Markov-chain
words of 4-32 symbols, analiticcl_amd/synth.py make_lexicon), 1.25 M of the 10 M length-bucketed queries of 4-32 symbols
(10 M / 8 GPUs), CLI defaults with max-edit-distance 2.  Size-independent properties (idempotence, shard == whole: the
multi-GPU split) and 600 spot checks against the C oracle built on the same 1 M-entry lexicon.  Reference: src/lib.rs:972-1027."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import synth
from oracle import cwrap as O

from fullsize_common import check_ranked, check_shards_equal_whole, checksum

NE = 1_000_000
NQ = 1_250_000


@pytest.fixture(scope="module")
def setup(data_dir, tmp_path_factory):
    words = list(dict.fromkeys(synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon")) +
                               synth.load_lexicon_words(os.path.join(data_dir, "nld.aspell.lexicon"))))
    lex = synth.make_lexicon(words, NE, seed=11)
    assert len(lex) == NE and len(set(lex)) == NE
    path = str(tmp_path_factory.mktemp("biglex") / "merged.lexicon")
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(lex) + "\n")
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(path)
    g.build()
    qs = synth.make_queries(lex, NQ, max_len=32, min_len=4, seed=5)
    qs.sort(key=len)  # length-bucketed
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
    b = g.encode_batch(qs, p)
    b.run()
    return g, path, qs, p, b, b.fetch_arrays(), b.stats()


def test_counts_and_idempotence(setup):
    g, _path, qs, p, b, (off, vid, dist, freq), st = setup
    assert g.num_instances() == NE
    assert st["n_queries"] == NQ and st["n_results"] == off[-1]
    assert 100 * NQ < st["n_pairs"] < 1000 * NQ
    lens = np.array([len(q) for q in qs])
    assert lens.min() >= 1 and lens.max() == 32 and (lens > 16).sum() > NQ // 8   # the 8-word and general kernels are in play
    c1 = checksum(off, vid, dist, freq)
    b.run()
    assert b.stats()["n_pairs"] == st["n_pairs"]
    assert checksum(*b.fetch_arrays()) == c1


def test_ranked_and_bounded(setup):
    _g, _path, _qs, _p, _b, (off, vid, dist, freq), _st = setup
    check_ranked(off, dist, NQ, 11, 0.25, 2.0)
    assert np.all(dist <= 1.0) and np.all(freq == 1.0)


def test_shards_equal_whole(setup):
    g, _path, qs, p, _b, arrays, _st = setup
    check_shards_equal_whole(g, qs, p, arrays, ((0, 30_000), (600_000, 640_000), (NQ - 25_000, NQ)))


def test_length_partitioned_share_has_fuller_tiles(setup, data_dir):
    """What a multi-device model gives ONE of 8 GPUs of the 10 M-query job (anx_model_to_devices: inputs ordered by length, cut into
    cost-balanced pieces): every query of the lengths the share owns.  Built here from the library's own split of a sample of the
    job (anx_debug_length_split) and the job's generator conditioned on length; its scan tiles hold more than twice the queries of
    the random eighth above (consecutive input ranges: an eighth of every (length, signature) group), rows checked against the oracle."""
    g, path, qs, p, _b, _arrays, st = setup
    lex = synth.load_lexicon_words(path)
    sample = synth.make_queries(lex, 400_000, max_len=32, min_len=4, seed=6)
    gid = g.length_split(sample, p, 8)
    counts = np.bincount(gid, minlength=8)
    assert counts.min() > 0 and counts.sum() == len(sample)          # balanced by COST: the cheap (long) lengths make shares of many queries
    lens = np.array([len(q.encode("utf-8")) for q in sample])      # the split orders by BYTE length
    for a, b2 in zip(range(7), range(1, 8)):                            # ascending lengths, neighbours share at most the boundary length
        assert lens[gid == a].max() <= lens[gid == b2].min()
    quota = {}
    for q, s in zip(sample, gid):
        if s == 4:
            quota[len(q)] = quota.get(len(q), 0) + 8                    # 3.2 M-query job: a 400 k share
    qs2 = synth.make_queries_with_quota(lex, quota, max_len=32, min_len=4, seed=7)
    b = g.encode_batch(qs2, p)
    b.run()
    st2 = b.stats()
    off, vid, dist, freq = b.fetch_arrays()
    b.free()
    fill_random = st["n_queries"] / st["n_scan_blocks"]
    fill_share = st2["n_queries"] / st2["n_scan_blocks"]
    assert len(qs2) < NQ / 2 and fill_share > 1.2 * fill_random, (fill_random, fill_share)   # a third of the queries, fuller tiles all the same
    o = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
    o.read_lexicon(path)
    o.build()
    op = O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0)
    idx = [int(i) for i in np.random.default_rng(4).choice(len(qs2), 200, replace=False)]
    _rc, res, cnts, _tp, _tc = o.find_variants_batch([qs2[i] for i in idx], op, nthreads=16, stride=16)
    for n, i in enumerate(idx):
        exp = [(res[n * 16 + j].vocab_id, res[n * 16 + j].dist_score, res[n * 16 + j].freq_score) for j in range(cnts[n])]
        assert [(int(vid[j]), float(dist[j]), float(freq[j])) for j in range(off[i], off[i + 1])] == exp, qs2[i]


def test_oracle_spot_check(setup, data_dir):
    g, path, qs, _p, _b, (off, vid, dist, freq), _st = setup
    o = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
    o.read_lexicon(path)
    o.build()
    op = O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0)
    rng = np.random.default_rng(3)
    # 20 000 of the 1.25 M queries (round 5: 600) through the oracle's OpenMP batch entry: ids in order, scores with ==
    idx = np.sort(rng.choice(NQ, 20_000, replace=False))
    c, ov, od, of, _tp, _tc = O.batch_rows(o, [qs[i] for i in idx], op, nthreads=16, stride=16)
    assert O.assert_rows_equal(off, vid, dist, freq, idx, c, ov, od, of, what=lambda i: qs[i]) > 20_000


def test_whole_job_equals_its_length_split_shares(setup):
    """The WHOLE configs[3] job shape on one GPU: 5 M of the 10 M length-bucketed queries (two device batches of 2.5 M), and the
    same 5 M queries as the 8 shares the length-partitioned split of a multi-device model makes of them (anx_debug_length_split: what
    anx_model_to_devices would give 8 GPUs), each share run on its own.  Rows of every query must agree -- a checksum of the whole
    run against the shares' rows scattered back to input order; independent inputs in any order: src/bin/analiticcl.rs:416-448."""
    g, path, _qs, p, _b, _arrays, _st = setup
    lex = synth.load_lexicon_words(path)
    NJ = 5_000_000
    job = synth.make_queries(lex, NJ, max_len=32, min_len=4, seed=9)
    whole_cnt = np.zeros(NJ, dtype=np.int64)
    parts = []
    for lo in range(0, NJ, 2_500_000):
        b = g.encode_batch(job[lo:lo + 2_500_000], p)
        b.run()
        off, vid, dist, freq = b.fetch_arrays()
        b.free()
        whole_cnt[lo:lo + 2_500_000] = np.diff(off)
        parts.append((vid, dist, freq))
    w_off = np.concatenate([[0], np.cumsum(whole_cnt)])
    w_vid, w_dist, w_freq = (np.concatenate([x[i] for x in parts]) for i in range(3))
    assert w_off[-1] > NJ    # more than a row per query on average
    gid = g.length_split(job, p, 8)
    assert np.bincount(gid, minlength=8).min() > 0
    s_cnt = np.zeros(NJ, dtype=np.int64)
    s_vid, s_dist, s_freq = np.zeros_like(w_vid), np.zeros_like(w_dist), np.zeros_like(w_freq)
    total_ms = []
    for sh in range(8):
        ix = np.nonzero(gid == sh)[0]
        b = g.encode_batch([job[i] for i in ix], p)
        b.run()
        total_ms.append(b.stats()["ms_total"])
        off, vid, dist, freq = b.fetch_arrays()
        b.free()
        cnt = np.diff(off)
        s_cnt[ix] = cnt
        assert np.array_equal(cnt, whole_cnt[ix])
        dst = np.repeat(w_off[ix], cnt) + (np.arange(off[-1]) - np.repeat(off[:-1], cnt))   # the rows of query ix[j] at its place in the whole job's arrays
        s_vid[dst], s_dist[dst], s_freq[dst] = vid, dist, freq
    assert np.array_equal(s_cnt, whole_cnt)
    assert checksum(w_off, s_vid, s_dist, s_freq) == checksum(w_off, w_vid, w_dist, w_freq)
    assert np.array_equal(s_vid, w_vid) and np.array_equal(s_dist, w_dist)
    print(f"whole job of {NJ} queries: shares' device times {[round(t, 2) for t in total_ms]} ms, balance {sum(total_ms) / 8 / max(total_ms):.2f}")
