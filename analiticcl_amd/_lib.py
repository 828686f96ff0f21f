"""Loader + ctypes signatures for libanx.so (the C ABI declared in include/anx.h).

The library is built in-tree (analiticcl_amd/libanx.so) by `python -m analiticcl_amd.build` or
`__graft_entry__.build()`.  There is no Python/CPU fallback: if the library is missing, importing fails
loudly; if no HIP device is present, calls that need the device raise AnxError(ANX_ENODEVICE).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ANX_LIB") or os.path.join(HERE, "libanx.so")  # ANX_LIB: an experiment build (tools/build_variant.sh)

ANX_OK, ANX_EINVAL, ANX_EIO, ANX_ENOTBUILT, ANX_ENODEVICE, ANX_ELIMIT, ANX_EEMPTY = 0, -1, -2, -3, -4, -5, -6
ANX_NO_VIA = 0xFFFFFFFFFFFFFFFF
ABI_VERSION = 3  # include/anx.h ANX_ABI_VERSION these ctypes signatures were written against (checked when the library is loaded)


class AnxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"anx error {code}: {message}")
        self.code = code


class Weights(C.Structure):
    _fields_ = [("ld", C.c_double), ("lcs", C.c_double), ("prefix", C.c_double), ("suffix", C.c_double),
                ("casew", C.c_double)]


class Threshold(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("value", C.c_uint8), ("ratio", C.c_float)]


class Params(C.Structure):
    _fields_ = [("max_anagram_distance", Threshold), ("max_edit_distance", Threshold),
                ("max_matches", C.c_uint64), ("score_threshold", C.c_double), ("cutoff_threshold", C.c_double),
                ("stop_at_exact_match", C.c_int32), ("freq_weight", C.c_float)]


class VocabParams(C.Structure):
    _fields_ = [("text_column", C.c_uint8), ("freq_column", C.c_int16), ("freq_handling", C.c_uint8),
                ("vocab_type", C.c_uint8)]


class Result(C.Structure):
    _fields_ = [("vocab_id", C.c_uint64), ("dist_score", C.c_double), ("freq_score", C.c_double),
                ("via", C.c_uint64)]


class SearchParams(C.Structure):
    _fields_ = [("base", Params), ("max_ngram", C.c_uint8), ("max_seq", C.c_uint32), ("lm_weight", C.c_float),
                ("variantmodel_weight", C.c_float), ("contextrules_weight", C.c_float), ("unicodeoffsets", C.c_int32)]


class Match(C.Structure):
    _fields_ = [("begin", C.c_size_t), ("end", C.c_size_t), ("n", C.c_uint32), ("selected", C.c_int32),
                ("var_begin", C.c_size_t), ("var_end", C.c_size_t), ("tag_begin", C.c_uint32),
                ("tag_end", C.c_uint32)]


class MatchTag(C.Structure):
    _fields_ = [("tag", C.c_uint16), ("seqnr", C.c_uint8), ("_pad", C.c_uint8)]


class Pair(C.Structure):
    _fields_ = [("query", C.c_uint32), ("vocab_id", C.c_uint32), ("ld", C.c_int16), ("lcs", C.c_uint16),
                ("prefixlen", C.c_uint16), ("suffixlen", C.c_uint16), ("samecase", C.c_uint8), ("_pad", C.c_uint8),
                ("score", C.c_double)]


class BatchStats(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("n_pairs", C.c_uint64), ("n_class_tests", C.c_uint64),
                ("n_results", C.c_uint64), ("n_scan_blocks", C.c_uint64), ("n_tests_kind", C.c_uint64 * 5),
                ("n_pair_slots", C.c_uint64), ("n_survivors", C.c_uint64), ("ms_scan", C.c_float),
                ("ms_group", C.c_float), ("ms_score", C.c_float), ("ms_rank", C.c_float), ("ms_total", C.c_float),
                ("ms_scan_kernel", C.c_float), ("n_selected", C.c_uint64), ("ms_filter_score_kernel", C.c_float),
                ("n_prefiltered_in_scan", C.c_uint64), ("n_conf_scripts", C.c_uint64), ("n_adj_tiles", C.c_uint64), ("n_adj_records", C.c_uint64), ("n_adj_records_first", C.c_uint64)]


_lib = None


def lib():
    """Load libanx.so once. torch (if installed) is imported first so that both share one HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -m analiticcl_amd.build` "
                          "(there is no CPU fallback for the variant-query path)")
    try:
        import torch  # noqa: F401  (loads the bundled libamdhip64 first; see DESIGN.md "Process model")
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    L.anx_abi_version.restype = C.c_int
    if L.anx_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {L.anx_abi_version()}, these bindings need {ABI_VERSION}: rebuild it "
                          "(`python -m analiticcl_amd.build`)")
    vp, cp, u64, sz = C.c_void_p, C.c_char_p, C.c_uint64, C.c_size_t
    sig = {
        "anx_last_error": (cp, []),
        "anx_abi_version": (C.c_int, []),
        "anx_last_error_code": (C.c_int, []),
        "anx_default_weights": (None, [C.POINTER(Weights)]),
        "anx_default_params": (None, [C.POINTER(Params)]),
        "anx_default_vocab_params": (None, [C.POINTER(VocabParams)]),
        "anx_model_new": (vp, [cp, C.POINTER(Weights), C.c_int]),
        "anx_model_new_with_alphabet": (vp, [cp, C.POINTER(Weights), C.c_int]),
        "anx_model_free": (None, [vp]),
        "anx_model_read_vocabulary": (C.c_int, [vp, cp, C.POINTER(VocabParams)]),
        "anx_model_add_to_vocabulary": (u64, [vp, cp, C.c_int, C.c_uint32, C.POINTER(VocabParams)]),
        "anx_model_add_variant": (C.c_int, [vp, u64, cp, C.c_double, C.c_int, C.c_uint32, C.POINTER(VocabParams)]),
        "anx_model_read_variants": (C.c_int, [vp, cp, C.POINTER(VocabParams), C.c_int]),
        "anx_model_add_to_confusables": (C.c_int, [vp, cp, C.c_double]),
        "anx_model_read_confusablelist": (C.c_int, [vp, cp]),
        "anx_model_set_confusables_before_pruning": (None, [vp]),
        "anx_edit_script": (C.c_int, [cp, cp, C.c_char_p, C.c_int]),
        "anx_model_confusable_weight": (C.c_int, [vp, cp, u64, C.POINTER(C.c_double)]),
        "anx_model_build": (C.c_int, [vp, C.c_int]),
        "anx_model_save_index": (C.c_int, [vp, cp]),
        "anx_model_load_index": (C.c_int, [vp, cp, C.c_int]),
        "anx_model_set_index_tag": (C.c_int, [vp, cp]),
        "anx_index_read_tag": (C.c_void_p, [cp]),
        "anx_model_num_lexicons": (u64, [vp]),
        "anx_model_lexicon_name": (cp, [vp, u64]),
        "anx_model_to_device": (C.c_int, [vp, C.c_int]),
        "anx_model_to_devices": (C.c_int, [vp, C.POINTER(C.c_int), C.c_int]),
        "anx_model_num_replicas": (C.c_int, [vp]),
        "anx_model_replica_device": (C.c_int, [vp, C.c_int]),
        "anx_debug_set_switch": (C.c_int, [cp, cp]),
        "anx_batch_num_shards": (C.c_int, [vp]),
        "anx_batch_shard_info": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(sz), C.POINTER(sz)]),
        "anx_batch_shard_inputs": (C.c_int, [vp, C.c_int, C.POINTER(C.POINTER(C.c_uint32))]),
        "anx_model_has": (C.c_int, [vp, cp]),
        "anx_model_vocab_size": (u64, [vp]),
        "anx_model_vocab_text": (cp, [vp, u64]),
        "anx_model_vocab_frequency": (C.c_uint32, [vp, u64]),
        "anx_model_vocab_lexindex": (C.c_uint32, [vp, u64]),
        "anx_model_num_instances": (u64, [vp]),
        "anx_model_num_classes": (u64, [vp]),
        "anx_model_bucket_size": (u64, [vp, C.c_int]),
        "anx_model_alphabet_size": (C.c_int, [vp]),
        "anx_model_normalize": (C.c_int, [vp, cp, C.c_char_p, C.c_int]),
        "anx_model_anahash": (C.c_int, [vp, cp, C.c_char_p, C.c_int]),
        "anx_find_variants_batch": (C.c_int, [vp, C.POINTER(cp), sz, C.POINTER(Params),
                                              C.POINTER(C.POINTER(Result)), C.POINTER(C.POINTER(sz))]),
        "anx_results_free": (None, [C.POINTER(Result), C.POINTER(sz)]),
        "anx_format_query_output": (C.c_int, [vp, C.POINTER(cp), sz, C.POINTER(Result), C.POINTER(sz), C.c_float, C.c_int,
                                              C.c_int, u64, C.POINTER(C.c_void_p), C.POINTER(sz)]),
        "anx_string_free": (None, [C.c_void_p]),
        "anx_format_search_output": (C.c_int, [vp, C.POINTER(cp), sz, C.POINTER(Match), C.POINTER(sz), C.POINTER(Result),
                                               C.POINTER(MatchTag), C.c_float, C.c_int, C.c_int, u64,
                                               C.POINTER(C.c_void_p), C.POINTER(sz)]),
        "anx_default_search_params": (None, [C.POINTER(SearchParams)]),
        "anx_find_all_matches_batch": (C.c_int, [vp, C.POINTER(cp), sz, C.POINTER(SearchParams),
                                                 C.POINTER(C.POINTER(Match)), C.POINTER(C.POINTER(sz)),
                                                 C.POINTER(C.POINTER(Result)), C.POINTER(sz),
                                                 C.POINTER(C.POINTER(MatchTag))]),
        "anx_matches_free": (None, [C.POINTER(Match), C.POINTER(sz), C.POINTER(Result), C.POINTER(MatchTag)]),
        "anx_model_add_contextrule": (C.c_int, [vp, cp, C.c_float, C.POINTER(cp), sz, C.POINTER(cp), sz]),
        "anx_model_read_contextrules": (C.c_int, [vp, cp]),
        "anx_model_num_tags": (sz, [vp]),
        "anx_model_tag_name": (cp, [vp, sz]),
        "anx_batch_encode": (vp, [vp, C.POINTER(cp), sz, C.POINTER(Params)]),
        "anx_batch_encode_packed": (vp, [vp, C.c_char_p, sz, sz, C.POINTER(Params)]),
        "anx_batch_gather_compact": (C.c_int, [vp, C.c_int, vp, sz, C.POINTER(sz), C.POINTER(sz)]),
        "anx_batch_encode_packed_device": (vp, [vp, vp, sz, sz, C.POINTER(Params)]),
        "anx_batch_encode_packed_device_on": (vp, [vp, vp, sz, sz, C.POINTER(Params), vp]),
        "anx_debug_search_stats": (C.c_int, [C.POINTER(C.c_uint64)]),
        "anx_debug_small_stats": (C.c_int, [C.POINTER(C.c_uint64)]),
        "anx_batch_run": (C.c_int, [vp, vp, vp]),
        "anx_batch_run_async": (C.c_int, [vp, vp, vp]),
        "anx_batch_wait": (C.c_int, [vp, vp]),
        "anx_batch_fetch": (C.c_int, [vp, C.POINTER(C.POINTER(Result)), C.POINTER(C.POINTER(sz))]),
        "anx_batch_fetch_compact": (C.c_int, [vp, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(C.c_uint32))]),
        "anx_compact_free": (None, [C.c_void_p, C.POINTER(C.c_uint32)]),
        "anx_compact_to_results": (None, [C.c_void_p, sz, C.POINTER(Result)]),
        "anx_batch_fetch_pairs": (C.c_int, [vp, C.POINTER(C.POINTER(Pair)), C.POINTER(sz)]),
        "anx_pairs_free": (None, [C.POINTER(Pair)]),
        "anx_batch_pair_counts": (C.c_int, [vp, C.POINTER(C.POINTER(C.c_uint32))]),
        "anx_counts_free": (None, [C.POINTER(C.c_uint32)]),
        "anx_batch_export_topk": (C.c_int, [vp, vp, C.c_uint32, vp]),
        "anx_batch_export_compact": (C.c_int, [vp, vp, sz, vp, C.POINTER(sz)]),
        "anx_batch_get_stats": (C.c_int, [vp, C.POINTER(BatchStats), C.c_size_t]),
        "anx_shutdown": (None, []),
        "anx_pipeline_new": (vp, [vp, C.c_int]),
        "anx_pipeline_submit_packed": (C.c_int, [vp, cp, C.c_size_t, C.c_size_t, C.POINTER(Params)]),
        "anx_pipeline_pending": (C.c_int, [vp]),
        "anx_pipeline_next": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_size_t)]),
        "anx_pipeline_free": (None, [vp]),
        "anx_debug_kernel_timer": (None, [C.c_int]),
        "anx_debug_kernel_time": (C.c_int, [cp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "anx_debug_length_split": (C.c_int, [vp, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(Params), C.c_int, vp, vp]),
        "anx_debug_signature": (C.c_int, [vp, cp, C.POINTER(C.c_uint64)]),
        "anx_debug_entries": (C.c_int, [vp, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_size_t)]),
        "anx_debug_adjacency": (C.c_int, [vp, C.c_int, C.c_uint64, vp, C.c_size_t, vp, C.POINTER(C.POINTER(C.c_uint32)), vp]),
        "anx_debug_adjacency_device": (C.c_int, [vp, vp, C.c_size_t, vp, C.POINTER(C.POINTER(C.c_uint32))]),
        "anx_debug_band_bound": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_size_t, C.c_int, C.c_int, vp]),
        "anx_batch_free": (None, [vp]),
        "anx_device_pool_trim": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED = None


def set_switch(name: str, value) -> None:
    """anx_debug_set_switch: the A/B and test switches are read from the environment once, when the library is first used;
    this changes one afterwards (value None = unset).  Tests use it instead of os.environ."""
    check(lib().anx_debug_set_switch(name.encode(), None if value is None else str(value).encode()))


def kernel_timer(enable: bool) -> None:
    """HIP-event timing of k_conf_script / k_lattice launches (anx_debug_kernel_timer): clears the totals."""
    lib().anx_debug_kernel_timer(1 if enable else 0)


def kernel_time(name: str):
    """(total ms, launches) of the timed launches of a kernel since kernel_timer(True); (0.0, 0) when none."""
    ms, n = C.c_double(0.0), C.c_uint64(0)
    lib().anx_debug_kernel_time(name.encode(), C.byref(ms), C.byref(n))
    return ms.value, n.value


def check(rc: int):
    if rc != 0:
        raise AnxError(rc, lib().anx_last_error().decode("utf-8", "replace"))


def last_error() -> str:
    return lib().anx_last_error().decode("utf-8", "replace")
