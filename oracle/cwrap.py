"""ctypes wrapper around oracle/libanx_oracle.so (TEST INFRASTRUCTURE -- see anx_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libanx_oracle.so")


class Threshold(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("value", C.c_uint8), ("ratio", C.c_float)]


class Params(C.Structure):
    _fields_ = [("max_anagram_distance", Threshold), ("max_edit_distance", Threshold),
                ("max_matches", C.c_uint64), ("score_threshold", C.c_double),
                ("cutoff_threshold", C.c_double), ("stop_at_exact_match", C.c_int32),
                ("freq_weight", C.c_float)]


class Result(C.Structure):
    _fields_ = [("vocab_id", C.c_uint64), ("dist_score", C.c_double), ("freq_score", C.c_double), ("via", C.c_uint64)]


class SearchParams(C.Structure):   # orc_search_params (anx_oracle.h): search mode of the C oracle (anx_oracle_search.inc)
    _fields_ = [("base", Params), ("max_ngram", C.c_uint8), ("max_seq", C.c_uint32), ("lm_weight", C.c_float),
                ("variantmodel_weight", C.c_float), ("contextrules_weight", C.c_float)]


class Match(C.Structure):          # orc_match
    _fields_ = [("begin", C.c_uint64), ("end", C.c_uint64), ("n", C.c_uint32), ("selected", C.c_int32),
                ("var_begin", C.c_uint64), ("var_end", C.c_uint64), ("has_variants", C.c_int32)]


class Pair(C.Structure):
    _fields_ = [("vocab_id", C.c_uint64), ("ld", C.c_int16), ("lcs", C.c_uint16),
                ("prefixlen", C.c_uint16), ("suffixlen", C.c_uint16), ("samecase", C.c_uint8)]


def build_lib(force: bool = False) -> str:
    srcs = [os.path.join(HERE, f) for f in ("anx_oracle.c", "anx_oracle_search.inc", "anx_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_lib()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.orc_model_new.restype = vp
        L.orc_model_new.argtypes = [C.c_char_p]
        L.orc_model_new_from_text.restype = vp
        L.orc_model_new_from_text.argtypes = [C.c_char_p]
        L.orc_model_free.argtypes = [vp]
        L.orc_set_weights.argtypes = [vp] + [C.c_double] * 5
        L.orc_alphabet_len.argtypes = [vp]
        L.orc_add.restype = C.c_uint64
        L.orc_add.argtypes = [vp, C.c_char_p, C.c_int, C.c_uint32]
        L.orc_read_lexicon.argtypes = [vp, C.c_char_p]
        L.orc_add_variant.argtypes = [vp, C.c_uint64, C.c_char_p, C.c_double, C.c_int, C.c_uint32, C.c_int]
        L.orc_read_variants.argtypes = [vp, C.c_char_p, C.c_int]
        L.orc_build.argtypes = [vp]
        L.orc_vocab_size.restype = C.c_uint64
        L.orc_vocab_size.argtypes = [vp]
        L.orc_vocab_text.restype = C.c_char_p
        L.orc_vocab_text.argtypes = [vp, C.c_uint64]
        L.orc_n_classes.restype = C.c_uint64
        L.orc_n_classes.argtypes = [vp]
        L.orc_n_instances.restype = C.c_uint64
        L.orc_n_instances.argtypes = [vp]
        L.orc_bucket_size.restype = C.c_uint64
        L.orc_bucket_size.argtypes = [vp, C.c_int]
        L.orc_has.argtypes = [vp, C.c_char_p]
        L.orc_anagram_instances.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_normalize.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_anahash_decimal.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_upper_bound.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_contains.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.orc_iter_parents.argtypes = [vp, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_iter_recursive.argtypes = [vp, C.c_char_p] + [C.c_int] * 8 + [C.c_char_p, C.c_int]
        for name in ("orc_damerau_levenshtein", "orc_levenshtein"):
            getattr(L, name).argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
        for name in ("orc_lcs", "orc_prefix", "orc_suffix"):
            getattr(L, name).argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_clamp_threshold.argtypes = [Threshold, C.c_int, C.c_int]
        L.orc_find_variants.argtypes = [vp, C.c_char_p, C.POINTER(Params), C.POINTER(Result), C.c_int,
                                        C.POINTER(Pair), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_find_nearest.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.orc_find_variants_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(Params), C.c_int,
                                              C.POINTER(Result), C.c_int, C.POINTER(C.c_int32),
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_add_lm.restype = C.c_uint64
        L.orc_add_lm.argtypes = [vp, C.c_char_p, C.c_int, C.c_uint32]
        L.orc_find_all_matches.argtypes = [vp, C.c_char_p, C.POINTER(SearchParams), C.POINTER(Match), C.c_int, C.POINTER(Result), C.c_int,
                                           C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        L.orc_find_all_matches_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(SearchParams), C.c_int, C.POINTER(C.c_int32),
                                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def threshold(th) -> Threshold:
    """("abs", x) | ("ratio", r) | ("ratiolimit", r, limit) -> Threshold"""
    if th[0] == "abs":
        return Threshold(0, int(th[1]), 0.0)
    if th[0] == "ratio":
        return Threshold(1, 0, float(th[1]))
    return Threshold(2, int(th[2]), float(th[1]))


def make_params(max_anagram_distance=("abs", 3), max_edit_distance=("abs", 3), max_matches=20,
                score_threshold=0.25, cutoff_threshold=2.0, stop_at_exact_match=False,
                freq_weight=0.0) -> Params:
    return Params(threshold(max_anagram_distance), threshold(max_edit_distance), max_matches,
                  score_threshold, cutoff_threshold, 1 if stop_at_exact_match else 0, freq_weight)


def make_search_params(base: Params, max_ngram=3, max_seq=250, lm_weight=1.0, variantmodel_weight=3.0, contextrules_weight=1.0) -> SearchParams:
    """SearchParameters' search-mode fields (src/types.rs:132-168; defaults as the reference's)"""
    return SearchParams(base, max_ngram, max_seq, lm_weight, variantmodel_weight, contextrules_weight)


def _b(s) -> bytes:
    return s if isinstance(s, bytes) else s.encode("utf-8")


class OracleModel:
    def __init__(self, alphabet_path: Optional[str] = None, alphabet_text: Optional[str] = None):
        L = lib()
        self.h = L.orc_model_new(_b(alphabet_path)) if alphabet_path else L.orc_model_new_from_text(_b(alphabet_text))
        if not self.h:
            raise RuntimeError(L.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_model_free(self.h)
            self.h = None

    def set_weights(self, ld, lcs, prefix, suffix, case):
        lib().orc_set_weights(self.h, ld, lcs, prefix, suffix, case)

    def add(self, text: str, freq: Optional[int] = None) -> int:
        return lib().orc_add(self.h, _b(text), 0 if freq is None else 1, freq or 0)

    def read_lexicon(self, path: str):
        if lib().orc_read_lexicon(self.h, _b(path)) != 0:
            raise RuntimeError(lib().orc_last_error().decode())

    def add_variant(self, ref_id: int, variant: str, score: float, freq: Optional[int] = None, transparent: bool = False):
        return bool(lib().orc_add_variant(self.h, ref_id, _b(variant), score, 0 if freq is None else 1, freq or 0,
                                          int(transparent)))

    def read_variants(self, path: str, transparent: bool = False):
        if lib().orc_read_variants(self.h, _b(path), int(transparent)) != 0:
            raise RuntimeError(lib().orc_last_error().decode())

    def build(self):
        lib().orc_build(self.h)

    def text(self, vid: int) -> str:
        return lib().orc_vocab_text(self.h, vid).decode("utf-8")

    def n_classes(self):
        return lib().orc_n_classes(self.h)

    def n_instances(self):
        return lib().orc_n_instances(self.h)

    def bucket_size(self, c):
        return lib().orc_bucket_size(self.h, c)

    def has(self, text):
        return bool(lib().orc_has(self.h, _b(text)))

    def anagram_instances(self, text) -> List[str]:
        buf = C.create_string_buffer(1 << 16)
        n = lib().orc_anagram_instances(self.h, _b(text), buf, len(buf))
        return buf.value.decode().split("\n")[:n]

    def normalize(self, text) -> List[int]:
        buf = C.create_string_buffer(256)
        n = lib().orc_normalize(self.h, _b(text), buf, 255)
        return list(buf.raw[:n])

    def anahash(self, text) -> int:
        buf = C.create_string_buffer(1024)
        if lib().orc_anahash_decimal(self.h, _b(text), buf, len(buf)) < 0:
            raise RuntimeError("anahash overflow")
        return int(buf.value)

    def upper_bound(self, text, alphabet_size) -> Tuple[int, int]:
        a, b = C.c_int(), C.c_int()
        lib().orc_upper_bound(self.h, _b(text), alphabet_size, C.byref(a), C.byref(b))
        return a.value, b.value

    def contains(self, a, b) -> bool:
        return bool(lib().orc_contains(self.h, _b(a), _b(b)))

    @staticmethod
    def _parse_nodes(buf, n):
        out = []
        for line in buf.value.decode().split("\n")[:n]:
            v, d, c = line.split(" ")
            out.append((int(v), int(d), int(c)))
        return out

    def iter_parents(self, text, alphabet_size):
        buf = C.create_string_buffer(1 << 20)
        n = lib().orc_iter_parents(self.h, _b(text), alphabet_size, buf, len(buf))
        return self._parse_nodes(buf, n)

    def iter_recursive(self, text, alphabet_size, singlebeam=False, mindepth=-1, maxdepth=-1, breadthfirst=False,
                       unique=False, empty_leaves=True, max_items=0):
        buf = C.create_string_buffer(1 << 22)
        n = lib().orc_iter_recursive(self.h, _b(text), alphabet_size, int(singlebeam), mindepth, maxdepth,
                                     int(breadthfirst), int(unique), int(empty_leaves), max_items, buf, len(buf))
        return self._parse_nodes(buf, n)

    def find_nearest(self, text, max_distance, stop_at_exact=False) -> List[int]:
        buf = C.create_string_buffer(1 << 24)
        n = lib().orc_find_nearest(self.h, _b(text), max_distance, int(stop_at_exact), buf, len(buf))
        if n < 0:
            raise RuntimeError("find_nearest overflow")
        return [int(x) for x in buf.value.decode().split("\n")[:n]]

    def find_variants_via(self, text, params: Params, cap: int = 1 << 14):
        """-> [(vocab_id, dist_score, freq_score, via | None)]"""
        res = (Result * cap)()
        npairs, ncls = C.c_int(0), C.c_int(0)
        n = lib().orc_find_variants(self.h, _b(text), C.byref(params), res, cap, None, C.byref(npairs), C.byref(ncls))
        if n < 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return [(res[i].vocab_id, res[i].dist_score, res[i].freq_score,
                 None if res[i].via == 0xFFFFFFFFFFFFFFFF else res[i].via) for i in range(n)]

    def find_variants(self, text, params: Params, want_pairs: bool = False, cap: int = 1 << 16):
        res = (Result * cap)()
        pair_cap = cap if want_pairs else 0
        pairs = (Pair * max(pair_cap, 1))()
        npairs = C.c_int(pair_cap)
        ncls = C.c_int(0)
        n = lib().orc_find_variants(self.h, _b(text), C.byref(params), res, cap,
                                    pairs if want_pairs else None, C.byref(npairs), C.byref(ncls))
        if n < 0:
            raise RuntimeError(lib().orc_last_error().decode())
        results = [(res[i].vocab_id, res[i].dist_score, res[i].freq_score) for i in range(n)]
        if want_pairs:
            pl = [(pairs[i].vocab_id, pairs[i].ld, pairs[i].lcs, pairs[i].prefixlen, pairs[i].suffixlen,
                   pairs[i].samecase) for i in range(min(npairs.value, pair_cap))]
            return results, pl, npairs.value, ncls.value
        return results

    def add_lm(self, text: str, freq: Optional[int] = None) -> int:
        """add_to_vocabulary(text, freq, VocabParams{vocab_type: LM})"""
        return lib().orc_add_lm(self.h, _b(text), 0 if freq is None else 1, 0 if freq is None else int(freq))

    def find_all_matches(self, text: str, sp: SearchParams):
        """-> [(matched text, begin, end, n, selected | None, variants [(vocab_id, dist, freq)] | None)] (byte offsets), scored pairs"""
        raw = _b(text)
        cap, rcap = len(raw) + 16, 32 * (len(raw) + 16)
        ms, rs = (Match * cap)(), (Result * rcap)()
        nr, npairs = C.c_int(0), C.c_uint64(0)
        n = lib().orc_find_all_matches(self.h, raw, C.byref(sp), ms, cap, rs, rcap, C.byref(nr), C.byref(npairs))
        if n < 0:
            raise RuntimeError("orc_find_all_matches: capacity")
        out = []
        for i in range(n):
            m = ms[i]
            var = [(rs[j].vocab_id, rs[j].dist_score, rs[j].freq_score) for j in range(m.var_begin, m.var_end)] if m.has_variants else None
            out.append((raw[m.begin:m.end].decode("utf-8"), m.begin, m.end, m.n, None if m.selected < 0 else m.selected, var))
        return out, npairs.value

    def find_all_matches_batch(self, texts: Sequence[str], sp: SearchParams, nthreads: int = 0):
        """the timed form (one OpenMP task per text) -> (rc, matches per text, total matches, total variant rows, scored pairs)"""
        n = len(texts)
        arr = (C.c_char_p * n)(*[_b(t) for t in texts])
        counts = (C.c_int32 * n)()
        tm, tr, tp = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = lib().orc_find_all_matches_batch(self.h, arr, n, C.byref(sp), nthreads, counts, C.byref(tm), C.byref(tr), C.byref(tp))
        return rc, list(counts), tm.value, tr.value, tp.value

    def find_variants_batch(self, texts: Sequence[str], params: Params, nthreads: int = 0, stride: int = 64):
        n = len(texts)
        arr = (C.c_char_p * n)(*[_b(t) for t in texts])
        res = (Result * (n * stride))()
        counts = (C.c_int32 * n)()
        tp, tc = C.c_uint64(), C.c_uint64()
        rc = lib().orc_find_variants_batch(self.h, arr, n, C.byref(params), nthreads, res, stride, counts,
                                           C.byref(tp), C.byref(tc))
        return rc, res, counts, tp.value, tc.value


def batch_rows(model: "OracleModel", texts: Sequence[str], params: Params, nthreads: int = 0, stride: int = 16):
    """orc_find_variants_batch as numpy arrays: (counts int32[n], vocab_id uint64[n, stride], dist f64[n, stride], freq f64[n, stride],
    scored pairs, anagram classes).  Rows beyond counts[i] are unspecified.  Raises if a list did not fit the stride."""
    import numpy as np
    rc, res, counts, tp, tc = model.find_variants_batch(texts, params, nthreads=nthreads, stride=stride)
    if rc != 0:
        raise RuntimeError(f"orc_find_variants_batch: {rc}")
    n = len(texts)
    dt = np.dtype([("vocab_id", "<u8"), ("dist", "<f8"), ("freq", "<f8"), ("via", "<u8")])
    a = np.frombuffer(res, dtype=dt).reshape(n, stride)
    c = np.frombuffer(counts, dtype=np.int32)
    if n and (c.min() < 0 or c.max() > stride):
        raise RuntimeError("a result list did not fit the stride")
    return c, a["vocab_id"], a["dist"], a["freq"], tp, tc


def assert_rows_equal(off, vid, dist, freq, idx, counts, ovid, odist, ofreq, what=lambda i: i, tol=0.0):
    """The ranked rows of the queries idx of a device result (CSR arrays off / vid / dist / freq) against batch_rows' arrays (row n =
    query idx[n]): list lengths, vocabulary ids in order, scores (== when tol is 0)."""
    import numpy as np
    idx = np.asarray(idx, dtype=np.int64)
    cnt = (off[idx + 1] - off[idx]).astype(np.int64)
    bad = np.nonzero(cnt != counts)[0]
    assert bad.size == 0, (what(int(idx[bad[0]])), int(cnt[bad[0]]), int(counts[bad[0]]))
    if cnt.sum() == 0:
        return 0
    src = np.repeat(off[idx], cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))   # device rows
    row = np.repeat(np.arange(len(idx)), cnt)
    col = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    for name, g, o in (("vocab_id", vid[src].astype(np.uint64), ovid[row, col]), ("dist_score", dist[src], odist[row, col]), ("freq_score", freq[src], ofreq[row, col])):
        ne = (g != o) if (tol == 0.0 or name == "vocab_id") else (np.abs(g - o) > tol)
        if ne.any():
            k = int(np.nonzero(ne)[0][0])
            raise AssertionError((name, what(int(idx[row[k]])), int(col[k]), g[k], o[k]))
    return int(cnt.sum())


def dl(s: Sequence[int], t: Sequence[int], maxd: int) -> Optional[int]:
    r = lib().orc_damerau_levenshtein(bytes(s), len(s), bytes(t), len(t), maxd)
    return None if r < 0 else r


def lev(s, t, maxd):
    r = lib().orc_levenshtein(bytes(s), len(s), bytes(t), len(t), maxd)
    return None if r < 0 else r


def lcs(s, t):
    return lib().orc_lcs(bytes(s), len(s), bytes(t), len(t))


def prefix(s, t):
    return lib().orc_prefix(bytes(s), len(s), bytes(t), len(t))


def suffix(s, t):
    return lib().orc_suffix(bytes(s), len(s), bytes(t), len(t))
