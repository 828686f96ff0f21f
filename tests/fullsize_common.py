"""Size-independent result checks shared by the full-size GPU tests (BASELINE.json configs[1..4])."""
import numpy as np


def checksum(off, vid, dist, freq) -> int:
    """Position-weighted XOR checksum over the CSR result arrays (order-sensitive inside every array)."""
    parts = (off.astype(np.uint64), vid.astype(np.uint64), dist.view(np.uint64), freq.view(np.uint64))
    acc = np.uint64(0)
    for i, p in enumerate(parts):
        w = (np.arange(p.size, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(i + 1)) | np.uint64(1)
        acc ^= np.bitwise_xor.reduce(p * w) if p.size else np.uint64(0)
    return int(acc ^ np.uint64(1469598103934665603))


def check_ranked(off, dist, n, max_rows, score_threshold, cutoff, score_floor_exact=True):
    """Rows of every query in descending dist_score order, at most max_rows of them, above the score threshold and (with
    a cutoff >= 1) above best / cutoff (src/lib.rs:1536-1622)."""
    cnt = np.diff(off)
    assert off.size == n + 1 and cnt.min() >= 0 and cnt.max() <= max_rows
    inner = np.ones(dist.size, dtype=bool)
    inner[off[:-1][cnt > 0]] = False
    assert np.all(dist[1:][inner[1:]] <= dist[:-1][inner[1:]])
    if score_floor_exact:
        assert np.all(dist >= score_threshold)
    if cutoff >= 1.0:
        best = np.zeros(n)
        best[cnt > 0] = dist[off[:-1][cnt > 0]]
        assert np.all(dist > np.repeat(best, cnt) / cutoff - 1e-15)
    return cnt


def check_shards_equal_whole(model, qs, params, arrays, ranges):
    """The multi-GPU split: a contiguous slice of the queries run on its own returns the rows of the whole run."""
    off, vid, dist, freq = arrays
    for lo, hi in ranges:
        b2 = model.encode_batch(qs[lo:hi], params)
        b2.run()
        o2, v2, d2, f2 = b2.fetch_arrays()
        b2.free()
        assert np.array_equal(o2, off[lo:hi + 1] - off[lo])
        sl = slice(off[lo], off[hi])
        assert np.array_equal(v2, vid[sl]) and np.array_equal(d2, dist[sl]) and np.array_equal(f2, freq[sl])
