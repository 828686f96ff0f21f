"""Python mirror of analiticcl's Python API for the variant-query path, on top of the anx C ABI.

Same class / method / keyword names as the reference's pyo3 binding
(/root/reference/bindings/python/src/lib.rs:591-812, typed in /root/reference/analiticcl.pyi):
`VariantModel`, `SearchParameters`, `Weights`, `VocabParams`; `find_variants`, `find_variants_par`.
All scoring runs on the GPU through libanx.so; nothing is computed in Python.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

from . import _lib as L


def _threshold(v) -> L.Threshold:
    """int -> Absolute, float in [0,1] -> Ratio, (float, int) -> RatioWithLimit
    (bindings/python/src/lib.rs extract_distance_threshold; src/types.rs:85-108)."""
    if isinstance(v, bool):
        raise ValueError("distance threshold must be int, float or (float, int)")
    if isinstance(v, int):
        if not 0 <= v <= 255:
            raise ValueError("absolute distance threshold must fit in u8")
        return L.Threshold(0, v, 0.0)
    if isinstance(v, float):
        if not 0.0 <= v <= 1.0:
            raise ValueError("ratio must be in range 0.0-1.0")
        return L.Threshold(1, 0, v)
    if isinstance(v, (tuple, list)) and len(v) == 2:
        return L.Threshold(2, int(v[1]), float(v[0]))
    raise ValueError("distance threshold must be int, float or (float, int)")


class Weights:
    """Weights of the score components (src/types.rs:40-73)."""

    _names = ("ld", "lcs", "prefix", "suffix", "case")

    def __init__(self, **kwargs):
        self.ld, self.lcs, self.prefix, self.suffix, self.case = 0.5, 0.125, 0.125, 0.125, 0.125
        for k, v in kwargs.items():
            if k not in self._names:
                raise ValueError(f"Unknown parameter for Weights: {k}")
            setattr(self, k, float(v))

    def _c(self) -> L.Weights:
        return L.Weights(self.ld, self.lcs, self.prefix, self.suffix, self.case)

    def to_dict(self) -> dict:
        return {k: getattr(self, k) for k in self._names}

    def get_ld(self): return self.ld
    def get_lcs(self): return self.lcs
    def get_prefix(self): return self.prefix
    def get_suffix(self): return self.suffix
    def get_case(self): return self.case
    def set_ld(self, value): self.ld = float(value)
    def set_lcs(self, value): self.lcs = float(value)
    def set_prefix(self, value): self.prefix = float(value)
    def set_suffix(self, value): self.suffix = float(value)
    def set_case(self, value): self.case = float(value)


class SearchParameters:
    """SearchParameters (src/types.rs:112-192); defaults are the library defaults (k=3, d=3, n=20)."""

    _defaults = dict(max_anagram_distance=3, max_edit_distance=3, max_matches=20, score_threshold=0.25,
                     cutoff_threshold=2.0, stop_criterion=False, max_ngram=3, lm_order=3, max_seq=250,
                     single_thread=False, context_weight=0.0, variantmodel_weight=3.0, lm_weight=1.0,
                     contextrules_weight=1.0, freq_weight=0.0, consolidate_matches=True, unicodeoffsets=False)

    def __init__(self, **kwargs):
        self.__dict__.update(self._defaults)
        for k, v in kwargs.items():
            if k not in self._defaults:
                raise ValueError(f"Unknown parameter for SearchParameters: {k}")
            setattr(self, k, v)

    def _c(self) -> L.Params:
        return L.Params(_threshold(self.max_anagram_distance), _threshold(self.max_edit_distance),
                        int(self.max_matches), float(self.score_threshold), float(self.cutoff_threshold),
                        1 if self.stop_criterion else 0, float(self.freq_weight))

    def _c_search(self) -> L.SearchParams:
        return L.SearchParams(self._c(), int(self.max_ngram), int(self.max_seq), float(self.lm_weight),
                              float(self.variantmodel_weight), float(self.contextrules_weight),
                              1 if self.unicodeoffsets else 0)

    def to_dict(self) -> dict:
        return {k: getattr(self, k) for k in self._defaults}

    # getters of the pyo3 class (bindings/python/src/lib.rs:261-345, analiticcl.pyi:75-137)
    @staticmethod
    def _threshold_value(v):
        """Absolute -> int, Ratio -> float, RatioWithLimit -> {"ratio", "limit"} (bindings/python/src/lib.rs:261-292)."""
        if isinstance(v, (tuple, list)):
            return {"ratio": float(v[0]), "limit": int(v[1])}
        return v

    def get_max_anagram_distance(self): return self._threshold_value(self.max_anagram_distance)
    def get_max_edit_distance(self): return self._threshold_value(self.max_edit_distance)
    def get_edit_distance(self): return self.get_max_edit_distance()  # the name analiticcl.pyi:81 documents
    def get_max_matches(self) -> int: return int(self.max_matches)
    def get_score_threshold(self) -> float: return float(self.score_threshold)
    def get_cutoff_threshold(self) -> float: return float(self.cutoff_threshold)
    def get_stop_criterion(self) -> bool: return bool(self.stop_criterion)
    def get_max_ngram(self) -> int: return int(self.max_ngram)
    def get_lm_order(self) -> int: return int(self.lm_order)
    def get_max_seq(self) -> int: return int(self.max_seq)
    def get_single_thread(self) -> bool: return bool(self.single_thread)
    def get_context_weight(self) -> float: return float(self.context_weight)
    def get_variantmodel_weight(self) -> float: return float(self.variantmodel_weight)
    def get_lm_weight(self) -> float: return float(self.lm_weight)
    def get_contextrules_weight(self) -> float: return float(self.contextrules_weight)
    def get_freq_weight(self) -> float: return float(self.freq_weight)
    def get_consolidate_matches(self) -> bool: return bool(self.consolidate_matches)
    def get_unicodeoffsets(self) -> bool: return bool(self.unicodeoffsets)


class VocabParams:
    """VocabParams (src/vocab.rs:108-143)."""

    _fh = {"sum": 0, "max": 1, "min": 2, "replace": 3}
    _vt = {"NONE": 0, "INDEXED": 1, "LM": 2, "TRANSPARENT": 4}

    def __init__(self, **kwargs):
        self.text_column, self.freq_column, self.freq_handling, self.vocabtype = 0, 1, "max", "INDEXED"
        for k, v in kwargs.items():
            if k not in ("text_column", "freq_column", "freq_handling", "vocabtype"):
                raise ValueError(f"Unknown parameter for VocabParams: {k}")
            setattr(self, k, v)

    def _c(self) -> L.VocabParams:
        vt = self.vocabtype
        vtv = vt if isinstance(vt, int) else sum(self._vt[x.strip().upper()] for x in vt.split("|"))
        return L.VocabParams(int(self.text_column), -1 if self.freq_column is None else int(self.freq_column),
                             self._fh[self.freq_handling.lower()], vtv)


def _b(s) -> bytes:
    return s if isinstance(s, bytes) else s.encode("utf-8")


def _pack(inputs: Sequence) -> bytes:
    """n inputs -> one buffer, each terminated by a NUL byte; an input with an embedded NUL is cut there, exactly as the
    char* entry points would."""
    if not len(inputs):
        return b""
    try:
        blob = ("\0".join(inputs) + "\0").encode("utf-8")          # all str: one join, one encode
    except TypeError:
        blob = b"\0".join(_b(t) for t in inputs) + b"\0"
    if blob.count(b"\0") != len(inputs):
        blob = b"\0".join(_b(t).split(b"\0", 1)[0] for t in inputs) + b"\0"
    return blob


def edit_script(source: str, target: str) -> str:
    """sesdiff::shortest_edit_script(source, target) in sesdiff notation (anx_edit_script)."""
    buf = C.create_string_buffer(4 * (len(_b(source)) + len(_b(target))) + 64)
    L.check(min(0, L.lib().anx_edit_script(_b(source), _b(target), buf, len(buf))))
    return buf.value.decode("utf-8")


def _result_columns(rows, total: int):
    """anx_result[total] -> (vocab_id, dist_score, freq_score, via) as Python lists, converted in bulk (element-wise
    ctypes access costs about a microsecond per field)."""
    import numpy as np
    if total == 0:
        return [], [], [], []
    dt = np.dtype([("vocab_id", "<u8"), ("dist", "<f8"), ("freq", "<f8"), ("via", "<u8")])
    a = np.frombuffer((C.c_char * (total * dt.itemsize)).from_address(C.addressof(rows.contents)), dtype=dt)
    return a["vocab_id"].tolist(), a["dist"].tolist(), a["freq"].tolist(), a["via"].tolist()


def _offsets(offs, n: int) -> List[int]:
    import numpy as np
    return np.ctypeslib.as_array(offs, shape=(n + 1,)).tolist()


class Batch:
    """A batch of queries encoded and resident in HBM (anx_batch_*): encode once, run many times."""

    def __init__(self, model: "VariantModel", inputs: Sequence[str], params: SearchParameters, packed: Optional[bytes] = None,
                 n: Optional[int] = None, device_ptr: Optional[int] = None, nbytes: int = 0, src_stream: Optional[int] = None):
        """inputs: the query strings; or packed + n: the same as ONE bytes object, every input followed by a NUL byte (what a
        caller that reads its queries from a file or a socket already has: the buffer goes to the device as it is)."""
        self.model = model
        cp = params._c()
        # one NUL-terminated buffer instead of a pointer array (anx_batch_encode_packed)
        if device_ptr is not None:   # the packed buffer already sits in HBM (anx_batch_encode_packed_device)
            self.n = int(n)
            if src_stream is None:   # the buffer is complete (its producer has been waited for)
                self.h = L.lib().anx_batch_encode_packed_device(model.h, C.c_void_p(device_ptr), int(nbytes), self.n, C.byref(cp))
            else:                    # ordered behind what the stream holds now (anx_batch_encode_packed_device_on)
                self.h = L.lib().anx_batch_encode_packed_device_on(model.h, C.c_void_p(device_ptr), int(nbytes), self.n, C.byref(cp), C.c_void_p(src_stream))
        else:
            blob = _pack(inputs) if packed is None else packed
            self.n = len(inputs) if packed is None else int(n)
            self.h = L.lib().anx_batch_encode_packed(model.h, blob, len(blob), self.n, C.byref(cp))
        if not self.h:
            raise L.AnxError(L.ANX_ENODEVICE if "device" in L.last_error() else L.ANX_EINVAL, L.last_error())
        self.freq_weight = float(params.freq_weight)

    def run(self, stream: int = 0):
        L.check(L.lib().anx_batch_run(self.model.h, self.h, C.c_void_p(stream)))

    def run_async(self, stream: int = 0):
        """Enqueue the run on `stream` and return (anx_batch_run_async); wait() completes it."""
        L.check(L.lib().anx_batch_run_async(self.model.h, self.h, C.c_void_p(stream)))

    def wait(self):
        L.check(L.lib().anx_batch_wait(self.model.h, self.h))

    def shards(self) -> List[tuple]:
        """[(device, first_input, n_inputs)]: the replicas of a multi-device model this batch is spread over."""
        out = []
        for g in range(L.lib().anx_batch_num_shards(self.h)):
            dev, lo, cnt = C.c_int(), C.c_size_t(), C.c_size_t()
            L.check(L.lib().anx_batch_shard_info(self.h, g, C.byref(dev), C.byref(lo), C.byref(cnt)))
            out.append((dev.value, lo.value, cnt.value))
        return out

    def shard_inputs(self, shard: int):
        """Indices (ascending) of the inputs shard `shard` holds under the length-partitioned split, None for a consecutive range."""
        import numpy as np
        dev, lo, cnt = C.c_int(), C.c_size_t(), C.c_size_t()
        L.check(L.lib().anx_batch_shard_info(self.h, shard, C.byref(dev), C.byref(lo), C.byref(cnt)))
        ix = C.POINTER(C.c_uint32)()
        L.check(L.lib().anx_batch_shard_inputs(self.h, shard, C.byref(ix)))
        return np.ctypeslib.as_array(ix, shape=(cnt.value,)).copy() if ix else None

    def stats(self) -> dict:
        s = L.BatchStats()
        L.check(L.lib().anx_batch_get_stats(self.h, C.byref(s), C.sizeof(s)))
        return {k: (list(getattr(s, k)) if k == "n_tests_kind" else getattr(s, k)) for k, _ in L.BatchStats._fields_}

    def fetch(self) -> List[List[tuple]]:
        """-> per query, ranked [(vocab_id, dist_score, freq_score)]"""
        rows = C.POINTER(L.Result)()
        offs = C.POINTER(C.c_size_t)()
        L.check(L.lib().anx_batch_fetch(self.h, C.byref(rows), C.byref(offs)))
        try:
            off = _offsets(offs, self.n)
            v, d, f, _via = _result_columns(rows, off[-1])
            return [list(zip(v[off[i]:off[i + 1]], d[off[i]:off[i + 1]], f[off[i]:off[i + 1]])) for i in range(self.n)]
        finally:
            L.lib().anx_results_free(rows, offs)

    def fetch_arrays(self):
        """-> (offsets[n+1], vocab_id[R], dist_score[R], freq_score[R]) as numpy arrays (CSR, batch order).  The three
        row arrays are strided VIEWS of the library's result buffer (32-byte anx_result records), which is released when
        the last of them is garbage collected: no second copy of the rows (141 MB per million queries of config 2)."""
        import weakref

        import numpy as np
        rows = C.POINTER(L.Result)()
        offs = C.POINTER(C.c_size_t)()
        L.check(L.lib().anx_batch_fetch(self.h, C.byref(rows), C.byref(offs)))
        release = True
        try:
            off = np.ctypeslib.as_array(offs, shape=(self.n + 1,)).astype(np.int64)
            total = int(off[-1])
            if total == 0:
                z = np.zeros(0)
                return off, z.astype(np.uint64), z, z
            dt = np.dtype([("vocab_id", "<u8"), ("dist", "<f8"), ("freq", "<f8"), ("via", "<u8")])
            owner = (C.c_char * (total * dt.itemsize)).from_address(C.addressof(rows.contents))
            weakref.finalize(owner, L.lib().anx_results_free, rows, offs)  # the views keep `owner` alive through .base
            release = False
            a = np.frombuffer(owner, dtype=dt)
            return off, a["vocab_id"], a["dist"], a["freq"]
        finally:
            if release:
                L.lib().anx_results_free(rows, offs)

    def fetch_compact(self):
        """-> (offsets[n+1] uint32, rows) with rows a structured array of 16-byte records (vocab_id u32, freq_score f32, dist_score
        f64): anx_batch_fetch_compact, half the bytes of fetch_arrays over PCIe.  Views of the library's pinned block, which is
        released when the last of them is garbage collected.  Not for models with variant lists or confusables."""
        import weakref

        import numpy as np
        rows = C.c_void_p()
        offs = C.POINTER(C.c_uint32)()
        L.check(L.lib().anx_batch_fetch_compact(self.h, C.byref(rows), C.byref(offs)))
        dt = np.dtype([("vocab_id", "<u4"), ("freq_score", "<f4"), ("dist_score", "<f8")])
        off_addr = C.addressof(offs.contents)
        # ONE owner for the whole block [rows | offsets]: both views keep it alive through .base
        owner = (C.c_char * (off_addr - rows.value + (self.n + 1) * 4)).from_address(rows.value)
        weakref.finalize(owner, L.lib().anx_compact_free, rows, offs)
        off = np.frombuffer(owner, dtype="<u4", count=self.n + 1, offset=off_addr - rows.value)
        return off, np.frombuffer(owner, dtype=dt, count=int(off[-1]))

    def fetch_pairs(self) -> List[tuple]:
        """-> every scored pair (query, vocab_id, ld|-1, lcs, prefixlen, suffixlen, samecase, score)"""
        pairs = C.POINTER(L.Pair)()
        n = C.c_size_t()
        L.check(L.lib().anx_batch_fetch_pairs(self.h, C.byref(pairs), C.byref(n)))
        try:
            return [(pairs[i].query, pairs[i].vocab_id, pairs[i].ld, pairs[i].lcs, pairs[i].prefixlen,
                     pairs[i].suffixlen, pairs[i].samecase, pairs[i].score) for i in range(n.value)]
        finally:
            L.lib().anx_pairs_free(pairs)

    def pair_counts(self):
        """Scored pairs per input, counted by the scan of a production run (anx_batch_pair_counts) -> numpy uint32[n]."""
        import numpy as np
        out = C.POINTER(C.c_uint32)()
        L.check(L.lib().anx_batch_pair_counts(self.h, C.byref(out)))
        try:
            return np.ctypeslib.as_array(out, shape=(max(self.n, 1),))[:self.n].copy()
        finally:
            L.lib().anx_counts_free(out)

    def export_topk(self, device_ptr: int, stride: int, stream: int = 0):
        L.check(L.lib().anx_batch_export_topk(self.h, C.c_void_p(device_ptr), stride, C.c_void_p(stream)))

    def export_compact(self, device_ptr: int, capacity: int, stream: int = 0) -> int:
        """offsets[n+1] (u32, padded to 16 bytes) + unpadded records into a device buffer; returns the bytes used."""
        used = C.c_size_t(0)
        L.check(L.lib().anx_batch_export_compact(self.h, C.c_void_p(device_ptr), capacity, C.c_void_p(stream), C.byref(used)))
        return used.value

    def gather_compact(self, dst_device: int, device_ptr: int, capacity: int):
        """anx_batch_gather_compact: every shard's compact export in one buffer on device dst_device (the shards' own devices copy
        their sections over) -> (section offsets [shards + 1], bytes used)."""
        import numpy as np
        ns = L.lib().anx_batch_num_shards(self.h)
        offs = (C.c_size_t * (ns + 1))()
        used = C.c_size_t(0)
        L.check(L.lib().anx_batch_gather_compact(self.h, dst_device, C.c_void_p(device_ptr), capacity, offs, C.byref(used)))
        return np.array(list(offs), dtype=np.int64), used.value

    def free(self):
        if getattr(self, "h", None):
            L.lib().anx_batch_free(self.h)
            self.h = None

    def __del__(self):
        self.free()


class Pipeline:
    """anx_pipeline: encode / run / fetch of consecutive packed batches overlapped for one caller thread.  submit() hands over a
    bytes object (every input followed by a NUL byte) and returns at once (or blocks while `depth` jobs are in flight); next()
    returns the oldest job's (offsets, rows) as Batch.fetch_compact does."""

    def __init__(self, model: "VariantModel", depth: int = 6):
        self.model = model
        self.h = L.lib().anx_pipeline_new(model.h, depth)
        if not self.h:
            raise L.AnxError(L.lib().anx_last_error_code(), L.lib().anx_last_error().decode("utf-8", "replace"))
        self._keep = []  # the submitted buffers stay alive until their results were returned

    def submit(self, packed: bytes, n: int, params: "SearchParameters"):
        """Raises AnxError (ANX_ELIMIT) when `depth` jobs are in flight: take a result with next() first."""
        L.check(L.lib().anx_pipeline_submit_packed(self.h, packed, len(packed), n, C.byref(params._c())))
        self._keep.append(packed)

    def pending(self) -> int:
        return L.lib().anx_pipeline_pending(self.h)

    def next(self):
        import weakref

        import numpy as np
        rows = C.c_void_p()
        offs = C.POINTER(C.c_uint32)()
        n = C.c_size_t()
        L.check(L.lib().anx_pipeline_next(self.h, C.byref(rows), C.byref(offs), C.byref(n)))
        self._keep.pop(0)
        dt = np.dtype([("vocab_id", "<u4"), ("freq_score", "<f4"), ("dist_score", "<f8")])
        off_addr = C.addressof(offs.contents)
        owner = (C.c_char * (off_addr - rows.value + (n.value + 1) * 4)).from_address(rows.value)
        weakref.finalize(owner, L.lib().anx_compact_free, rows, offs)
        off = np.frombuffer(owner, dtype="<u4", count=n.value + 1, offset=off_addr - rows.value)
        return off, np.frombuffer(owner, dtype=dt, count=int(off[-1]))

    def close(self):
        if self.h:
            L.lib().anx_pipeline_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class VariantModel:
    """VariantModel (src/lib.rs:50-100) for the query path; `device` = HIP device ordinal
    (default: $LOCAL_RANK or 0; -1 = host index only).  `devices` = a list of ordinals: one process drives a replica of the
    lexicon on each of them (anx_model_to_devices) and every batch call shards its inputs over the replicas."""

    def __init__(self, alphabet_file: str, weights: Optional[Weights] = None, debug: int = 0,
                 device: Optional[int] = None, alphabet_text: Optional[str] = None, devices: Optional[List[int]] = None):
        w = (weights or Weights())._c()
        lib = L.lib()
        if alphabet_text is not None:
            self.h = lib.anx_model_new_with_alphabet(_b(alphabet_text), C.byref(w), debug)
        else:
            self.h = lib.anx_model_new(_b(alphabet_file), C.byref(w), debug)
        if not self.h:
            raise L.AnxError(L.ANX_EIO, L.last_error())
        self.devices = list(devices) if devices else None
        self.device = self.devices[0] if self.devices else (int(os.environ.get("LOCAL_RANK", "0")) if device is None else device)
        self.lexicons: List[str] = []

    def __del__(self):
        try:
            if getattr(self, "h", None):
                L.lib().anx_model_free(self.h)
                self.h = None
        except Exception:  # interpreter shutdown: the module globals are already gone
            pass

    # -- loading -----------------------------------------------------------------------------------
    def read_vocabulary(self, filename: str, params: Optional[VocabParams] = None):
        self.__dict__.pop("_vocab_cache", None)
        p = (params or VocabParams())._c()
        L.check(L.lib().anx_model_read_vocabulary(self.h, _b(filename), C.byref(p)))
        self.lexicons.append(filename)

    def read_lexicon(self, filename: str):
        self.__dict__.pop("_vocab_cache", None)
        self.read_vocabulary(filename, VocabParams())

    def add_to_vocabulary(self, text: str, frequency: Optional[int] = None, params: Optional[VocabParams] = None):
        self.__dict__.pop("_vocab_cache", None)
        p = (params or VocabParams())._c()
        return L.lib().anx_model_add_to_vocabulary(self.h, _b(text), 0 if frequency is None else 1,
                                                   frequency or 0, C.byref(p))

    def add_variant(self, ref_id: int, variant: str, score: float, frequency: Optional[int] = None,
                    params: Optional[VocabParams] = None) -> bool:
        """add_variant (src/lib.rs:460): link `variant` to the reference item `ref_id` with a score"""
        self.__dict__.pop("_vocab_cache", None)
        p = (params or VocabParams())._c()
        rc = L.lib().anx_model_add_variant(self.h, ref_id, _b(variant), float(score), 0 if frequency is None else 1,
                                           frequency or 0, C.byref(p))
        if rc < 0:
            L.check(rc)
        return bool(rc)

    def read_variants(self, filename: str, transparent: bool = False):
        """Load a weighted variant list; transparent=True for error lists whose items are never returned themselves
        (bindings/python/src/lib.rs:671-681)"""
        self.__dict__.pop("_vocab_cache", None)
        p = VocabParams()._c()
        L.check(L.lib().anx_model_read_variants(self.h, _b(filename), C.byref(p), 1 if transparent else 0))
        self.lexicons.append(filename)

    def build(self):
        self.__dict__.pop("_vocab_cache", None)
        if self.devices and len(self.devices) > 1:
            L.check(L.lib().anx_model_build(self.h, -1))
            self.to_devices(self.devices)
        else:
            L.check(L.lib().anx_model_build(self.h, self.device))

    def set_index_tag(self, tag: str):
        """Stored in the image save_index writes: what the model was built from (see index_tag_of)."""
        L.check(L.lib().anx_model_set_index_tag(self.h, _b(tag)))

    @staticmethod
    def index_tag_of(filename: str) -> Optional[str]:
        """The tag of an index image without loading it; None if the file is not an image of this library version."""
        p = L.lib().anx_index_read_tag(_b(filename))
        if not p:
            return None
        try:
            return C.string_at(p).decode("utf-8", "replace")
        finally:
            L.lib().anx_string_free(p)

    def save_index(self, filename: str):
        """Write the built model (vocabulary + the lexicon image the GPU consumes) to disk (anx_model_save_index)."""
        L.check(L.lib().anx_model_save_index(self.h, _b(filename)))

    def load_index(self, filename: str):
        """Instead of read_lexicon / read_variants / build: load an image written by save_index for the same alphabet."""
        self.__dict__.pop("_vocab_cache", None)
        multi = bool(self.devices and len(self.devices) > 1)
        L.check(L.lib().anx_model_load_index(self.h, _b(filename), -1 if multi else self.device))
        if multi:
            self.to_devices(self.devices)
        n = L.lib().anx_model_num_lexicons(self.h) if hasattr(L.lib(), "anx_model_num_lexicons") else 0
        self.lexicons = [L.lib().anx_model_lexicon_name(self.h, i).decode("utf-8") for i in range(n)]

    def to_device(self, device: int):
        self.device = device
        self.devices = None
        L.check(L.lib().anx_model_to_device(self.h, device))

    def to_devices(self, devices: List[int]):
        """One replica of the built lexicon per listed HIP device (an ordinal may repeat); batch calls then shard their inputs
        over the replicas inside this one process (anx_model_to_devices)."""
        arr = (C.c_int * len(devices))(*devices)
        L.check(L.lib().anx_model_to_devices(self.h, arr, len(devices)))
        self.devices = list(devices)
        self.device = self.devices[0]

    @property
    def num_replicas(self) -> int:
        return L.lib().anx_model_num_replicas(self.h)

    # -- introspection ------------------------------------------------------------------------------
    def __contains__(self, text: str) -> bool:
        return bool(L.lib().anx_model_has(self.h, _b(text)))

    def vocab_text(self, vocab_id: int) -> str:
        return L.lib().anx_model_vocab_text(self.h, vocab_id).decode("utf-8")

    def num_instances(self) -> int:
        return L.lib().anx_model_num_instances(self.h)

    def num_classes(self) -> int:
        return L.lib().anx_model_num_classes(self.h)

    def bucket_size(self, charcount: int) -> int:
        return L.lib().anx_model_bucket_size(self.h, charcount)

    def normalize(self, text: str) -> List[int]:
        buf = C.create_string_buffer(256)
        n = L.lib().anx_model_normalize(self.h, _b(text), buf, 255)
        if n < 0:
            L.check(n)
        return list(buf.raw[:n])

    def anahash(self, text: str) -> int:
        buf = C.create_string_buffer(4096)
        n = L.lib().anx_model_anahash(self.h, _b(text), buf, len(buf))
        if n < 0:
            L.check(n)
        return int(buf.value)

    def length_split(self, inputs: Sequence[str], params: "SearchParameters", n_shards: int, learn_ms=None):
        """Which of n_shards replicas each input would go to under the length-partitioned split of a multi-device model
        (anx_debug_length_split; needs no device): a numpy uint8 array.  learn_ms: measured device times of those shares, fed to
        the split's cost corrections (the next call then returns the corrected split)."""
        import numpy as np
        arr = (C.c_char_p * len(inputs))(*[_b(t) for t in inputs])
        out = np.zeros(len(inputs), dtype=np.uint8)
        ms = np.ascontiguousarray(learn_ms, dtype=np.float64) if learn_ms is not None else None
        L.check(L.lib().anx_debug_length_split(self.h, arr, len(inputs), C.byref(params._c()), n_shards, out.ctypes.data,
                                               ms.ctypes.data if ms is not None else None))
        return out

    # -- the hot path ---------------------------------------------------------------------------------
    def encode_packed(self, packed: bytes, n: int, params: SearchParameters) -> Batch:
        """encode_batch for n inputs already packed into one bytes object, each followed by a NUL byte."""
        return Batch(self, (), params, packed=packed, n=n)

    def encode_packed_device(self, device_ptr: int, nbytes: int, n: int, params: SearchParameters, stream: Optional[int] = None) -> Batch:
        """encode_packed for a packed buffer in device memory (e.g. a torch uint8 tensor's data_ptr()): nothing crosses PCIe.
        stream = None: the buffer must be complete (synchronised) when the call is made; stream = a hipStream_t handle (0 = the default
        stream): the encoder is ordered behind what that stream holds now (the kernel or copy that fills the buffer)."""
        return Batch(self, (), params, n=n, device_ptr=device_ptr, nbytes=nbytes, src_stream=stream)

    def encode_batch(self, inputs: Sequence[str], params: SearchParameters) -> Batch:
        return Batch(self, inputs, params)

    def find_variants_ids(self, inputs: Sequence[str], params: SearchParameters, with_via: bool = False
                          ) -> List[List[tuple]]:
        """anx_find_variants_batch: -> per input, ranked [(vocab_id, dist_score, freq_score[, via | None])]"""
        n = len(inputs)
        arr = (C.c_char_p * max(n, 1))(*[_b(t) for t in inputs])
        cp = params._c()
        rows = C.POINTER(L.Result)()
        offs = C.POINTER(C.c_size_t)()
        L.check(L.lib().anx_find_variants_batch(self.h, arr, n, C.byref(cp), C.byref(rows), C.byref(offs)))
        try:
            off = _offsets(offs, n)
            v, d, f, via = _result_columns(rows, off[-1])
            if with_via:
                via = [None if x == L.ANX_NO_VIA else x for x in via]
                return [list(zip(v[off[i]:off[i + 1]], d[off[i]:off[i + 1]], f[off[i]:off[i + 1]], via[off[i]:off[i + 1]]))
                        for i in range(n)]
            return [list(zip(v[off[i]:off[i + 1]], d[off[i]:off[i + 1]], f[off[i]:off[i + 1]])) for i in range(n)]
        finally:
            L.lib().anx_results_free(rows, offs)

    def query_output(self, inputs: Sequence[str], params: SearchParameters, json: bool = False,
                     output_lexmatch: bool = False, first_seqnr: int = 1) -> str:
        """One device batch + the text `analiticcl query` prints for it (anx_format_query_output: the TSV lines or JSON
        items of src/bin/analiticcl.rs:21-187, formatted natively)."""
        n = len(inputs)
        arr = (C.c_char_p * max(n, 1))(*[_b(t) for t in inputs])
        cp = params._c()
        rows = C.POINTER(L.Result)()
        offs = C.POINTER(C.c_size_t)()
        L.check(L.lib().anx_find_variants_batch(self.h, arr, n, C.byref(cp), C.byref(rows), C.byref(offs)))
        try:
            buf, ln = C.c_void_p(), C.c_size_t(0)
            L.check(L.lib().anx_format_query_output(self.h, arr, n, rows, offs, float(params.freq_weight), int(bool(json)),
                                                    int(bool(output_lexmatch)), first_seqnr, C.byref(buf), C.byref(ln)))
            try:
                return C.string_at(buf, ln.value).decode("utf-8")
            finally:
                L.lib().anx_string_free(buf)
        finally:
            L.lib().anx_results_free(rows, offs)

    def _to_dict(self, vid: int, dist: float, freq: float, freq_weight: float, via: Optional[int] = None) -> Dict:
        # variantresult_to_dict, bindings/python/src/lib.rs:554-588
        fw = float(freq_weight)
        score = dist if fw == 0.0 else (dist + fw * freq) / (1.0 + fw)
        cache = self.__dict__.setdefault("_vocab_cache", {})  # vocabulary items do not change after build()
        hit = cache.get(vid)
        if hit is None:
            lexindex = L.lib().anx_model_vocab_lexindex(self.h, vid)
            hit = cache[vid] = (self.vocab_text(vid), [name for i, name in enumerate(self.lexicons) if lexindex & (1 << i)])
        d = {"text": hit[0], "score": score, "dist_score": dist, "freq_score": freq}
        if via is not None:
            d["via"] = self.vocab_text(via)
        d["lexicons"] = list(hit[1])
        return d

    def find_variants(self, input: str, params: SearchParameters) -> List[dict]:
        res = self.find_variants_ids([input], params, with_via=True)[0]
        return [self._to_dict(v, d, f, params.freq_weight, via) for v, d, f, via in res]

    def find_variants_par(self, input: List[str], params: SearchParameters) -> List[dict]:
        res = self.find_variants_ids(input, params, with_via=True)
        return [{"input": t, "variants": [self._to_dict(v, d, f, params.freq_weight, via) for v, d, f, via in r]}
                for t, r in zip(input, res)]

    # -- search mode: the caller of the hot path ---------------------------
    def find_all_matches_ids(self, texts: Sequence[str], params: SearchParameters) -> List[List[dict]]:
        """anx_find_all_matches_batch over many texts: every segment of one n-gram order, over all texts, is one
        device batch. Per text: [{begin, end, n, selected, variants: [(vocab_id, dist, freq, via|None)]}]."""
        n = len(texts)
        enc = [_b(t) for t in texts]
        arr = (C.c_char_p * max(n, 1))(*enc)
        sp = params._c_search()
        ms, offs, rows = C.POINTER(L.Match)(), C.POINTER(C.c_size_t)(), C.POINTER(L.Result)()
        tags = C.POINTER(L.MatchTag)()
        nrows = C.c_size_t(0)
        L.check(L.lib().anx_find_all_matches_batch(self.h, arr, n, C.byref(sp), C.byref(ms), C.byref(offs),
                                                   C.byref(rows), C.byref(nrows), C.byref(tags)))
        try:
            import numpy as np
            off = _offsets(offs, n)
            v, d, f, via = _result_columns(rows, nrows.value)
            via = [None if x == L.ANX_NO_VIA else x for x in via]
            out = [[] for _ in range(n)]
            if off[-1]:
                mdt = np.dtype([("begin", "<u8"), ("end", "<u8"), ("n", "<u4"), ("selected", "<i4"), ("vb", "<u8"), ("ve", "<u8"),
                                ("tb", "<u4"), ("te", "<u4")])
                ma = np.frombuffer((C.c_char * (off[-1] * mdt.itemsize)).from_address(C.addressof(ms.contents)), dtype=mdt)
                cols = [ma[k].tolist() for k in ("begin", "end", "n", "selected", "vb", "ve", "tb", "te")]
                ntags = max(cols[7]) if cols[7] else 0
                tg = sq = []
                if ntags:
                    tdt = np.dtype([("tag", "<u2"), ("seqnr", "u1"), ("pad", "u1")])
                    ta = np.frombuffer((C.c_char * (ntags * tdt.itemsize)).from_address(C.addressof(tags.contents)), dtype=tdt)
                    tg, sq = ta["tag"].tolist(), ta["seqnr"].tolist()
                for i in range(n):
                    cur = out[i]
                    for j in range(off[i], off[i + 1]):
                        vb, ve, tb, te = cols[4][j], cols[5][j], cols[6][j], cols[7][j]
                        cur.append({"begin": cols[0][j], "end": cols[1][j], "n": cols[2][j],
                                    "selected": None if cols[3][j] < 0 else cols[3][j],
                                    "variants": list(zip(v[vb:ve], d[vb:ve], f[vb:ve], via[vb:ve])),
                                    "tag": tg[tb:te], "seqnr": sq[tb:te]})
            return out
        finally:
            L.lib().anx_matches_free(ms, offs, rows, tags)

    def find_all_matches_arrays(self, texts: Sequence[str], params: SearchParameters):
        """Bulk form of find_all_matches_ids: numpy copies of what anx_find_all_matches_batch returns, no per-match Python
        objects.  -> (offs[n+1], matches (structured: begin, end, n, selected, vb, ve, tb, te), rows (structured:
        vocab_id, dist, freq, via)); the matches of text i are matches[offs[i]:offs[i+1]], the variants of a match
        rows[vb:ve] (selected = index into them, -1 = none)."""
        import numpy as np
        n = len(texts)
        arr = (C.c_char_p * max(n, 1))(*[_b(t) for t in texts])
        sp = params._c_search()
        ms, offs, rows = C.POINTER(L.Match)(), C.POINTER(C.c_size_t)(), C.POINTER(L.Result)()
        nrows = C.c_size_t(0)
        L.check(L.lib().anx_find_all_matches_batch(self.h, arr, n, C.byref(sp), C.byref(ms), C.byref(offs),
                                                   C.byref(rows), C.byref(nrows), None))
        try:
            off = np.ctypeslib.as_array(offs, shape=(n + 1,)).astype(np.int64)
            mdt = np.dtype([("begin", "<u8"), ("end", "<u8"), ("n", "<u4"), ("selected", "<i4"), ("vb", "<u8"), ("ve", "<u8"),
                            ("tb", "<u4"), ("te", "<u4")])
            rdt = np.dtype([("vocab_id", "<u8"), ("dist", "<f8"), ("freq", "<f8"), ("via", "<u8")])
            nm = int(off[-1])
            ma = np.frombuffer((C.c_char * (nm * mdt.itemsize)).from_address(C.addressof(ms.contents)), dtype=mdt).copy() \
                if nm else np.zeros(0, dtype=mdt)
            ra = np.frombuffer((C.c_char * (nrows.value * rdt.itemsize)).from_address(C.addressof(rows.contents)), dtype=rdt).copy() \
                if nrows.value else np.zeros(0, dtype=rdt)
            return off, ma, ra
        finally:
            L.lib().anx_matches_free(ms, offs, rows, None)

    def search_output(self, texts: Sequence[str], params: SearchParameters, json: bool = False,
                      output_lexmatch: bool = False, first_seqnr: int = 1):
        """find_all_matches over the texts + the text `analiticcl search` prints for the matches
        (anx_format_search_output) -> (text, number of matches)."""
        if params.unicodeoffsets:
            raise ValueError("search_output needs byte offsets (unicodeoffsets=False)")
        n = len(texts)
        arr = (C.c_char_p * max(n, 1))(*[_b(t) for t in texts])
        sp = params._c_search()
        ms, offs, rows = C.POINTER(L.Match)(), C.POINTER(C.c_size_t)(), C.POINTER(L.Result)()
        tags = C.POINTER(L.MatchTag)()
        nrows = C.c_size_t(0)
        L.check(L.lib().anx_find_all_matches_batch(self.h, arr, n, C.byref(sp), C.byref(ms), C.byref(offs),
                                                   C.byref(rows), C.byref(nrows), C.byref(tags)))
        try:
            buf, ln = C.c_void_p(), C.c_size_t(0)
            L.check(L.lib().anx_format_search_output(self.h, arr, n, ms, offs, rows, tags, float(params.freq_weight),
                                                     int(bool(json)), int(bool(output_lexmatch)), first_seqnr,
                                                     C.byref(buf), C.byref(ln)))
            try:
                return C.string_at(buf, ln.value).decode("utf-8"), int(offs[n])
            finally:
                L.lib().anx_string_free(buf)
        finally:
            L.lib().anx_matches_free(ms, offs, rows, tags)

    def find_all_matches(self, text: str, params: SearchParameters) -> List[dict]:
        """find_all_matches of the pyo3 binding (bindings/python/src/lib.rs:752-805): the selected variant first."""
        res = self.find_all_matches_ids([text], params)[0]
        raw = _b(text)
        out = []
        for m in res:
            inp = text[m["begin"]:m["end"]] if params.unicodeoffsets else raw[m["begin"]:m["end"]].decode("utf-8")
            order = list(range(len(m["variants"])))
            if m["selected"] is not None and m["selected"] < len(order):
                order.remove(m["selected"])
                order.insert(0, m["selected"])
            item = {"input": inp, "offset": {"begin": m["begin"], "end": m["end"]}}
            if m["tag"]:  # bindings/python/src/lib.rs:768-782
                item["tag"] = [self.tag_name(t) for t in m["tag"]]
                item["seqnr"] = list(m["seqnr"])
            item["variants"] = [self._to_dict(*m["variants"][k][:3], params.freq_weight, m["variants"][k][3])
                                for k in order]
            out.append(item)
        return out

    # -- context rules of search mode (src/lib.rs:570-765; bindings/python/src/lib.rs:630-700) -------------
    def add_contextrule(self, pattern: str, score: float, tag: Sequence[str] = (), tagoffset: Sequence[str] = ()):
        t = (C.c_char_p * max(1, len(tag)))(*[_b(x) for x in tag])
        o = (C.c_char_p * max(1, len(tagoffset)))(*[_b(x) for x in tagoffset])
        L.check(L.lib().anx_model_add_contextrule(self.h, _b(pattern), float(score), t, len(tag), o, len(tagoffset)))

    def read_contextrules(self, filename: str):
        L.check(L.lib().anx_model_read_contextrules(self.h, _b(filename)))

    @property
    def tags(self) -> List[str]:
        return [self.tag_name(i) for i in range(L.lib().anx_model_num_tags(self.h))]

    def tag_name(self, index: int) -> str:
        s = L.lib().anx_model_tag_name(self.h, index)
        if s is None:
            raise IndexError("tag %d" % index)
        return s.decode("utf-8")

    # -- confusables: host-side rescoring of the ranked lists ---------------
    def read_lm(self, filename: str):
        """Language-model n-gram counts: read_vocabulary with VocabType::LM (bindings/python/src/lib.rs:659-667)."""
        self.read_vocabulary(filename, VocabParams(vocabtype="LM"))

    def read_confusiblelist(self, filename: str):
        """The spelling the reference's Python API uses (analiticcl.pyi:283, bindings/python/src/lib.rs read_confusiblelist)."""
        self.read_confusablelist(filename)

    def read_confusablelist(self, filename: str):
        L.check(L.lib().anx_model_read_confusablelist(self.h, _b(filename)))

    def add_to_confusables(self, editscript: str, weight: float):
        L.check(L.lib().anx_model_add_to_confusables(self.h, _b(editscript), float(weight)))

    def compute_confusable_weight(self, input: str, vocab_id: int) -> float:
        """compute_confusable_weight (src/lib.rs:1733-1756): product of the weights of the confusable patterns found in the edit
        script input -> vocabulary item."""
        w = C.c_double(1.0)
        L.check(L.lib().anx_model_confusable_weight(self.h, _b(input), int(vocab_id), C.byref(w)))
        return w.value

    def set_confusables_before_pruning(self):
        L.lib().anx_model_set_confusables_before_pruning(self.h)
