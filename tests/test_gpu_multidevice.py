"""The multi-GPU split behind the C ABI (the reference's one-process fan-out over inputs,
src/bin/analiticcl.rs:445-448, src/lib.rs:1883): anx_model_to_devices replicates the lexicon, every batch call shards its
inputs over the replicas (one host thread + stream per replica) and returns the rows concatenated in input order.  The GPU boxes
have ONE device, so the replicas here are {0, 0} / {0, 0, 0}: two or three copies of the lexicon on the one GPU, driven
concurrently -- the code path of N GPUs except for the device ordinal.  shard == whole (rows, scores, pair counts, statistics)
on the shapes of BASELINE configs[1], [2] (confusables) and [3] (long queries, StopAtExactMatch), the staged and the one-shot
forms, variant lists, and search mode."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import synth

CONF = os.path.join(synth.GOLDEN_DATA, "confusables10.tsv")


@pytest.fixture(autouse=True)
def _small_shards():
    A.set_switch("ANX_SHARD_MIN", 64)   # the default (8192 inputs per replica) would keep the small cases on one replica
    yield
    A.set_switch("ANX_SHARD_MIN", None)
    A.set_switch("ANX_SHARD_POLICY", None)


def _check_shards(b_shards, b_inputs, qs, devices, policy):
    """range: consecutive input ranges in input order.  length (the default): every replica holds whole byte lengths -- only a length
    a cut runs through is shared by two neighbouring replicas -- and the shards partition the inputs."""
    n = len(qs)
    assert [s[0] for s in b_shards] == devices[:len(b_shards)] and sum(s[2] for s in b_shards) == n
    if policy == "range":
        assert all(ix is None for ix in b_inputs)
        assert all(b_shards[i][1] + b_shards[i][2] == b_shards[i + 1][1] for i in range(len(b_shards) - 1))
        return
    lens = np.array([len(q.encode("utf-8")) for q in qs])
    seen = np.zeros(n, dtype=bool)
    ranges = []
    for (dev, lo, cnt), ix in zip(b_shards, b_inputs):
        assert ix is not None and ix.size == cnt and ix[0] == lo and np.all(np.diff(ix.astype(np.int64)) > 0) and not seen[ix].any()
        seen[ix] = True
        ranges.append((int(np.minimum(lens[ix], 255).min()), int(np.minimum(lens[ix], 255).max())))
    assert seen.all()
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 <= b0 and a0 <= b0 and a1 <= b1, ranges   # ascending length ranges that overlap in at most the boundary length


def _model(data_dir, lex, devices, confusables=False, variants=None):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), devices=devices)
    g.read_lexicon(os.path.join(data_dir, f"{lex}.aspell.lexicon"))
    if confusables:
        g.read_confusablelist(CONF)
    if variants:
        g.read_variants(variants)
    g.build()
    return g


def _run(g, qs, p, packed=False, counts=True, compact=False):
    if packed:
        blob = b"".join(q.encode("utf-8") + b"\0" for q in qs)
        b = g.encode_packed(blob, len(qs), p)
    else:
        b = g.encode_batch(qs, p)
    b.run_async()
    b.wait()
    out = dict(arrays=b.fetch_arrays(), stats=b.stats(), shards=b.shards(), inputs=[b.shard_inputs(g_) for g_ in range(len(b.shards()))])
    if compact:   # the 16-byte records over the shards == the anx_result rows
        off, vid, dist, freq = out["arrays"]
        coff, crows = b.fetch_compact()
        assert np.array_equal(coff, off) and np.array_equal(crows["vocab_id"], vid) and np.array_equal(crows["dist_score"], dist)
        assert np.array_equal(crows["freq_score"], freq.astype(np.float32))
    if counts:
        out["counts"] = b.pair_counts()
    b.run()  # a second run of a sharded batch (buffers sized from the first)
    again = b.fetch_arrays()
    for x, y in zip(out["arrays"], again):
        assert np.array_equal(x, y)
    b.free()
    return out


def _same(a, b, counts=True):
    for x, y in zip(a["arrays"], b["arrays"]):
        assert np.array_equal(x, y)
    if counts:
        assert np.array_equal(a["counts"], b["counts"])
    for k in ("n_queries", "n_pairs", "n_results", "n_survivors"):
        assert a["stats"][k] == b["stats"][k], k


@pytest.mark.parametrize("lex,n,max_len,kw,conf", [
    ("eng", 200_000, 16, dict(max_anagram_distance=3, max_edit_distance=2, max_matches=10), False),          # configs[1] shape
    ("nld", 120_000, 24, dict(max_anagram_distance=3, max_edit_distance=3, max_matches=10), True),           # configs[2]: confusables
    ("nld", 60_000, 32, dict(max_anagram_distance=3, max_edit_distance=2, max_matches=5, stop_criterion=True, freq_weight=0.5), False),
])
def test_two_and_three_replicas_equal_one(data_dir, lex, n, max_len, kw, conf):
    words = synth.load_lexicon_words(os.path.join(data_dir, f"{lex}.aspell.lexicon"))
    qs = synth.make_queries(words, n, max_len=max_len, seed=synth.SEED + 7)
    qs[5] = ""            # inputs without results in the middle of a shard and at the shard edges
    qs[n // 2 - 1] = ""
    qs[n // 2] = "x" * 300
    p = A.SearchParameters(**kw)
    one = _run(_model(data_dir, lex, [0], conf), qs, p)
    assert len(one["shards"]) == 1
    for devices in ([0, 0], [0, 0, 0]):
        g = _model(data_dir, lex, devices, conf)
        assert g.num_replicas == len(devices)
        for policy in ("length", "range"):
            A.set_switch("ANX_SHARD_POLICY", policy)
            for packed in (False, True):
                got = _run(g, qs, p, packed=packed, compact=not conf)
                _check_shards(got["shards"], got["inputs"], qs, devices, policy)
                _same(one, got)
        A.set_switch("ANX_SHARD_POLICY", None)
        # the one-shot call (anx_find_variants_batch) over the replicas
        ids = g.find_variants_ids(qs[:30_000], p)
        off, vid, dist, freq = one["arrays"]
        for i in (0, 5, 77, 14_999, 15_000, 29_999):
            assert ids[i] == [(int(vid[j]), float(dist[j]), float(freq[j])) for j in range(off[i], off[i + 1])]
        del g


def test_uneven_byte_split_and_tail(data_dir):
    """The packed form splits by BYTES: a buffer whose second half holds much shorter strings gives the replicas different
    input counts; strings behind the announced n are ignored; more announced than present is an error."""
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    qs = synth.make_queries(words, 20_000, max_len=30, min_len=12, seed=5) + synth.make_queries(words, 60_000, max_len=4, seed=6)
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    one = _run(_model(data_dir, "eng", [0]), qs, p)
    g = _model(data_dir, "eng", [0, 0])
    blob = b"".join(q.encode("utf-8") + b"\0" for q in qs)
    for policy in ("range", "length"):
        A.set_switch("ANX_SHARD_POLICY", policy)
        got = _run(g, qs, p, packed=True)
        assert got["shards"][0][2] != got["shards"][1][2]
        _same(one, got)
        b = g.encode_packed(blob, 50_001, p)   # the announced n cuts inside the second shard (range) / drops the tail (length)
        b.run()
        off, vid, _d, _f = b.fetch_arrays()
        assert sum(s[2] for s in b.shards()) == 50_001
        b.free()
        assert np.array_equal(off, one["arrays"][0][:50_002]) and np.array_equal(vid, one["arrays"][1][:off[-1]])
    A.set_switch("ANX_SHARD_POLICY", None)
    with pytest.raises(A.AnxError, match="fewer strings than announced"):
        g.encode_packed(blob, len(qs) + 1, p)
    with pytest.raises(A.AnxError, match="own stream"):
        b = g.encode_batch(qs, p)
        try:
            b.run(stream=0x1234)
        finally:
            b.free()


def test_small_calls_use_one_replica_and_exports_need_one_shard(data_dir):
    import torch
    A.set_switch("ANX_SHARD_MIN", None)
    g = _model(data_dir, "eng", [0, 0])
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    qs = synth.make_queries(words, 20_000, max_len=16, seed=9)
    b = g.encode_batch(qs[:5000], p)     # below 2 x 8192: one shard, exports work
    assert len(b.shards()) == 1
    b.run()
    buf = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:0")
    assert b.export_compact(buf.data_ptr(), buf.numel()) > 0
    torch.cuda.synchronize()
    b.free()
    b = g.encode_batch(qs, p)
    assert len(b.shards()) == 2
    b.run()
    with pytest.raises(A.AnxError, match="several replicas"):
        b.export_compact(buf.data_ptr(), buf.numel())
    b.free()


def test_variant_lists_over_replicas(data_dir, tmp_path):
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    vl = tmp_path / "variants.tsv"
    rng = np.random.default_rng(3)
    refs = [w for w in words[::97] if w.isalpha() and len(w) > 4][:400]
    with open(vl, "w", encoding="utf-8") as f:
        for w in refs:
            i = int(rng.integers(1, len(w) - 1))
            f.write(f"{w}\t{w[:i] + w[i + 1:]}\t0.9\t{w[:i] + w[i] + w[i:]}\t0.8\n")
    qs = synth.make_queries(refs, 40_000, max_len=20, seed=11)
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    one = _run(_model(data_dir, "eng", [0], variants=str(vl)), qs, p, counts=False)
    got = _run(_model(data_dir, "eng", [0, 0], variants=str(vl)), qs, p, counts=False)
    _same(one, got, counts=False)


def test_search_mode_over_replicas(data_dir):
    """find_all_matches drives the same staged calls: every n-gram order's segment batch is sharded over the replicas."""
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    texts = synth.make_running_text(words, 0.4, seed=21)
    sp = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, max_ngram=3)
    one = _model(data_dir, "eng", [0]).find_all_matches_arrays(texts, sp)
    two = _model(data_dir, "eng", [0, 0]).find_all_matches_arrays(texts, sp)
    for x, y in zip(one, two):
        assert np.array_equal(x, y)
    assert one[1].size > 10_000


def test_conf_fallback_with_an_empty_shard(data_dir):
    """Confusables weighted on the device, a string beyond the device kernel's working memory (> 64 code points) -> the whole batch is
    redone with the host weighting, which downloads the inputs from every shard -- also from a shard that received NO input (range
    policy, packed form: the byte-balanced cut falls inside the long last string)."""
    words = synth.load_lexicon_words(os.path.join(data_dir, "nld.aspell.lexicon"))
    qs = synth.make_queries(words, 6, max_len=12, seed=3) + ["x" * 40 + "y" * 700]
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    A.set_switch("ANX_SHARD_MIN", 1)
    one = _run(_model(data_dir, "nld", [0], True), qs, p, counts=False)
    g = _model(data_dir, "nld", [0, 0], True)
    for policy in ("range", "length"):
        A.set_switch("ANX_SHARD_POLICY", policy)
        for packed in (True, False):
            got = _run(g, qs, p, packed=packed, counts=False)
            for x, y in zip(one["arrays"], got["arrays"]):
                assert np.array_equal(x, y)
    A.set_switch("ANX_SHARD_POLICY", None)


@pytest.mark.parametrize("policy", ["length", "range"])
def test_gather_compact_behind_the_c_abi(data_dir, policy):
    """anx_batch_gather_compact: the compact top-k export of every shard of a three-replica batch (replicas on device 0: the pool's boxes
    have one GPU; a shard on another device would come over by hipMemcpyPeerAsync) in ONE device buffer -- what analiticcl_amd/shard.py
    gathers with RCCL for one rank per GPU.  Every section (u32 offsets + 16-byte records, in the shard's input order) must hold the rows
    anx_batch_fetch_compact returns for those inputs; a buffer that is too small is refused with the size it needs."""
    import torch
    from analiticcl_amd import shard as SH
    A.set_switch("ANX_SHARD_MIN", "1")
    A.set_switch("ANX_SHARD_POLICY", policy)
    try:
        g = _model(data_dir, "eng", [0, 0, 0])
        p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
        words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
        qs = synth.make_queries(words, 60_000, max_len=16, seed=19)
        b = g.encode_batch(qs, p)
        assert len(b.shards()) == 3
        b.run()
        off, rows = b.fetch_compact()
        small = torch.empty(1024, dtype=torch.uint8, device="cuda:0")
        with pytest.raises(A.AnxError, match="too small"):
            b.gather_compact(0, small.data_ptr(), small.numel())
        buf = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:0")
        so, used = b.gather_compact(0, buf.data_ptr(), buf.numel())
        torch.cuda.synchronize()
        raw = buf[:used].cpu().numpy().tobytes()
        assert so[-1] == used and len(so) == 4
        seen = np.zeros(len(qs), dtype=bool)
        for s, (_dev, lo, cnt) in enumerate(b.shards()):
            ix = b.shard_inputs(s)
            ix = np.arange(lo, lo + cnt) if ix is None else ix
            o = np.frombuffer(raw, dtype="<u4", count=cnt + 1, offset=int(so[s]))
            r = np.frombuffer(raw, dtype=SH.TOPK_DTYPE, count=int(o[cnt]), offset=int(so[s]) + SH.compact_offsets_bytes(cnt))
            assert np.array_equal(np.diff(o).astype(np.int64), (off[ix + 1].astype(np.int64) - off[ix].astype(np.int64)))
            want = np.concatenate([rows[off[i]:off[i + 1]] for i in ix]) if len(ix) else rows[:0]
            assert np.array_equal(r["vocab_id"], want["vocab_id"]) and np.array_equal(r["dist_score"], want["dist_score"])
            seen[ix] = True
        assert seen.all()
        b.free()
    finally:
        A.set_switch("ANX_SHARD_MIN", None)
        A.set_switch("ANX_SHARD_POLICY", None)
