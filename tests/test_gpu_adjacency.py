"""The signature adjacency lists BUILT ON THE DEVICE (analiticcl_amd/csrc/adjacency.hip: keys by sort + unique, one wave per list
through the signature hash table) against the host builder (adjacency.cpp, itself checked against brute force on the CPU by
tests/test_adjacency_cpu.py): the same signatures get lists, every list has the same row counts per length section and the same
entries per section (the order inside a section is the builder's own), padding included.  And the run with host-built lists
(ANX_ADJ_BUILD=host) returns the same rows."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import analiticcl_amd as A
from analiticcl_amd import _lib as L
from analiticcl_amd import synth


def _host_lists(m, closure, sigs):
    cum = np.zeros((len(sigs), 8), dtype=np.uint32)
    ids = C.POINTER(C.c_uint32)()
    stats = (C.c_uint64 * 7)()
    L.check(L.lib().anx_debug_adjacency(m.h, closure, 1 << 42, sigs.ctypes.data_as(C.c_void_p), len(sigs), cum.ctypes.data_as(C.c_void_p), C.byref(ids), stats))
    return cum, ids, list(stats)


def _device_lists(m, sigs):
    cum = np.zeros((len(sigs), 8), dtype=np.uint32)
    ids = C.POINTER(C.c_uint32)()
    L.check(L.lib().anx_debug_adjacency_device(m.h, sigs.ctypes.data_as(C.c_void_p), len(sigs), cum.ctypes.data_as(C.c_void_p), C.byref(ids)))
    return cum, ids


@pytest.mark.parametrize("lex,closure", [("eng", 2), ("nld", 1), ("eng", 0)])
def test_device_built_lists_equal_the_host_builder(data_dir, lex, closure):
    A.set_switch("ANX_ADJ_CLOSURE", str(closure))
    try:
        m = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
        m.read_lexicon(os.path.join(data_dir, f"{lex}.aspell.lexicon"))
        m.build()
    finally:
        A.set_switch("ANX_ADJ_CLOSURE", None)
    words = synth.load_lexicon_words(os.path.join(data_dir, f"{lex}.aspell.lexicon"))
    rng = np.random.default_rng(3)
    # signatures of lexicon words, of perturbed words (the closure and beyond) and far-away ones
    texts = list(rng.choice(words, 400)) + synth.make_queries(words, 1200, max_len=24, seed=5) + ["zzzzqqqqxxxxjjjj", "a", "q" * 40]
    sig = np.zeros(len(texts), dtype=np.uint64)
    s = C.c_uint64()
    for i, t in enumerate(texts):
        L.check(L.lib().anx_debug_signature(m.h, t.encode(), C.byref(s)))
        sig[i] = s.value
    sig = np.unique(sig)
    hcum, hids, hstats = _host_lists(m, closure, sig)
    dcum, dids = _device_lists(m, sig)
    try:
        assert np.array_equal(hcum[:, 1:], dcum[:, 1:])                      # the same signatures have lists, same rows per section
        have = hcum[:, 0] != 0xFFFFFFFF
        assert have.sum() > 300 and (~have).sum() >= 1 and np.array_equal(have, dcum[:, 0] != 0xFFFFFFFF)
        nent = hstats[3] and None
        for i in np.nonzero(have)[0]:
            for sct in range(7):
                r0h, r0d = int(hcum[i, 0]) + (int(hcum[i, sct]) if sct else 0), int(dcum[i, 0]) + (int(dcum[i, sct]) if sct else 0)
                rows = int(hcum[i, sct + 1]) - (int(hcum[i, sct]) if sct else 0)
                a = np.ctypeslib.as_array(hids, shape=((r0h + rows) * 64 + 1,))[r0h * 64:(r0h + rows) * 64]
                b = np.ctypeslib.as_array(dids, shape=((r0d + rows) * 64 + 1,))[r0d * 64:(r0d + rows) * 64]
                assert np.array_equal(np.sort(a), np.sort(b)), (i, sct)
    finally:
        C.CDLL(None).free(hids)
        C.CDLL(None).free(dids)


def test_host_built_lists_give_the_same_rows(data_dir):
    words = synth.load_lexicon_words(os.path.join(data_dir, "eng.aspell.lexicon"))
    qs = synth.make_queries(words, 150_000, max_len=20, seed=8)
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10)
    out = {}
    for mode in ("device", "host"):
        A.set_switch("ANX_ADJ_BUILD", "host" if mode == "host" else None)
        try:
            m = A.VariantModel(os.path.join(data_dir, "simple.alphabet.tsv"), A.Weights(), device=0)
            m.read_lexicon(os.path.join(data_dir, "eng.aspell.lexicon"))
            m.build()
        finally:
            A.set_switch("ANX_ADJ_BUILD", None)
        b = m.encode_batch(qs, p)
        b.run()
        st = b.stats()
        out[mode] = (st["n_pairs"], st["n_survivors"], st["n_adj_tiles"], st["n_adj_records"]) + tuple(x.tobytes() for x in b.fetch_arrays())
        b.free()
    assert out["device"] == out["host"] and out["device"][2] > 0
