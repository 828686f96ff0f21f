"""Round-6 GPU tests: the paths a caller uses (fresh batches through anx_pipeline, inputs already in HBM) against the ORACLE rather than
against the synchronous path, a replica that loads without its adjacency lists, search mode's early output with more parts than
workers, the stream-ordered device encoder.  Reference: src/lib.rs:972-1027 (find_variants), src/bin/analiticcl.rs:416-448 (fan-out)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import _lib as L
from analiticcl_amd import synth
from oracle import cwrap as O


@pytest.fixture(scope="module")
def eng(data_dir):
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    g.build()
    o = O.OracleModel(alphabet_path=os.path.join(data_dir, "simple.alphabet.tsv"))
    o.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
    o.build()
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    return g, o, words


P = dict(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
OP = (("abs", 3), ("abs", 2), 10, 0.25, 2.0)


def _compact_vs_oracle(o, qs, off, rows, nsample, seed):
    idx = np.sort(np.random.default_rng(seed).choice(len(qs), min(nsample, len(qs)), replace=False))
    c, ov, od, of, _tp, _tc = O.batch_rows(o, [qs[i] for i in idx], O.make_params(*OP), nthreads=16, stride=16)
    # compact records carry freq_score as f32 (exact for the 1.0 of a lexicon without frequencies)
    return O.assert_rows_equal(off.astype(np.int64), rows["vocab_id"], rows["dist_score"], rows["freq_score"].astype(np.float64), idx, c, ov, od, of,
                               what=lambda i: qs[i])


def test_pipeline_batches_vs_oracle(eng):
    """anx_pipeline (encode / launch / wait / fetch of consecutive fresh batches on the library's threads and streams): rows of two jobs
    in flight together against the oracle -- not against the synchronous path."""
    g, o, words = eng
    p = A.SearchParameters(**P)
    sets = [synth.make_queries(words, n, max_len=16, seed=700 + i) for i, n in enumerate((200_000, 150_000, 60_000))]
    blobs = [("\0".join(qs) + "\0").encode("utf-8") for qs in sets]
    pl = A.Pipeline(g, depth=3)
    for blob, qs in zip(blobs, sets):
        pl.submit(blob, len(qs), p)
    got = [pl.next() for _ in sets]
    pl.close()
    for k, ((off, rows), qs) in enumerate(zip(got, sets)):
        assert off.size == len(qs) + 1
        assert _compact_vs_oracle(o, qs, off, rows, 6000, 10 + k) > 6000


def test_encode_packed_device_vs_oracle_and_stream_order(eng):
    """anx_batch_encode_packed_device (inputs already in HBM) against the oracle; anx_batch_encode_packed_device_on: the buffer is FILLED
    by work enqueued on a side stream that has not been synchronised when the encoder is called (a long chain of copies first, then the
    real bytes): the encoder's stream must wait for it."""
    import torch
    g, o, words = eng
    p = A.SearchParameters(**P)
    qs = synth.make_queries(words, 300_000, max_len=16, seed=42)
    blob = ("\0".join(qs) + "\0").encode("utf-8")
    src = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    b = g.encode_packed_device(src.data_ptr(), src.numel(), len(qs), p)
    b.run()
    off, rows = b.fetch_compact()
    ref = (off.copy(), rows.copy())
    b.free()
    assert _compact_vs_oracle(o, qs, ref[0], ref[1], 8000, 3) > 8000
    # stream-ordered form: `dst` holds garbage (every byte 'z', no NUL at all) until the side stream's last copy lands
    side = torch.cuda.Stream()
    dst = torch.full((src.numel(),), ord("z"), dtype=torch.uint8, device="cuda")
    big = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(24):            # ~10 ms of device work ahead of the copy that makes the buffer valid
            big.add_(1)
        dst.copy_(src, non_blocking=True)
    b2 = g.encode_packed_device(dst.data_ptr(), dst.numel(), len(qs), p, stream=side.cuda_stream)
    b2.run()
    off2, rows2 = b2.fetch_compact()
    b2.free()
    assert np.array_equal(off2, ref[0]) and np.array_equal(rows2, ref[1])


def test_adjacency_failure_leaves_a_working_replica(data_dir, eng):
    """a failed build of the signature adjacency lists (out of memory on a busy device) must not fail the model load --
    the lists only accelerate the scan.  ANX_ADJ_FAIL injects the failure after the build's allocations."""
    _g, o, words = eng
    A.set_switch("ANX_ADJ_FAIL", "1")
    try:
        g2 = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
        g2.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
        g2.build()
    finally:
        A.set_switch("ANX_ADJ_FAIL", None)
    qs = synth.make_queries(words, 40_000, max_len=16, seed=9)
    p = A.SearchParameters(**P)
    b = g2.encode_batch(qs, p)
    b.run()
    st = b.stats()
    off, rows = b.fetch_compact()
    b.free()
    assert st["n_adj_tiles"] == 0 and st["n_scan_blocks"] > 0
    assert _compact_vs_oracle(o, qs, off, rows, 3000, 1) > 3000


def test_early_output_stays_on_with_more_parts_than_workers(data_dir):
    """every worker zeroed the upper bound its part had published when the part finished; with more parts than workers
    the sum came out too small and the early output was silently dropped for exactly the large calls it was written for."""
    lex = os.path.join(data_dir, "eng.aspell.lexicon")
    words = synth.load_lexicon_words(lex)
    common = [w for w in words if w.isalpha()][::29][:3000]
    g = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
    g.read_lexicon(lex)
    g.build()
    texts = synth.make_running_text(common, 1.5, seed=3)
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, max_ngram=2)

    def stats():
        out = (C.c_uint64 * 4)()
        assert L.lib().anx_debug_search_stats(out) == 0
        return list(out)
    A.set_switch("ANX_SEARCH_PARTS_MIN", "1")
    A.set_switch("ANX_SEARCH_PART_BYTES", str(128 << 10))   # ~12 parts for 1.5 MB ...
    A.set_switch("ANX_SEARCH_PARTS", "2")                   # ... on 2 workers
    try:
        s0 = stats()
        off, ma, ra = g.find_all_matches_arrays(texts, p)
        s1 = stats()
        assert s1[0] == s0[0] + 1, (s0, s1)           # it ran as several parts
        assert s1[1] == s0[1] + 1 and s1[2] == s0[2]  # ... and kept its early output
        A.set_switch("ANX_SEARCH_PARTS", "1")
        A.set_switch("ANX_SEARCH_PARTS_MIN", str(1 << 30))
        off1, ma1, ra1 = g.find_all_matches_arrays(texts, p)
        assert np.array_equal(off, off1) and np.array_equal(ma, ma1) and np.array_equal(ra, ra1)
    finally:
        for sw in ("ANX_SEARCH_PARTS_MIN", "ANX_SEARCH_PART_BYTES", "ANX_SEARCH_PARTS"):
            A.set_switch(sw, None)

