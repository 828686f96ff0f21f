"""The signature adjacency lists (analiticcl_amd/csrc/adjacency.h) against brute force, on the CPU.

A list must hold exactly the index entries whose group-sum signature lies within L1 distance 3 of the list's signature
(a superset of find_nearest_anahashes' candidates for any k <= 3, /root/reference/src/lib.rs:1143-1308: the scan's exact per-class
test decides the rest), every entry once, in the section of its length, sections padded to whole rows of 64 with the padding id.
"""
import ctypes as C

import numpy as np
import pytest

import analiticcl_amd as A
from analiticcl_amd import _lib as L


def _sig_bytes(sigs):
    return np.asarray(sigs, dtype=np.uint64).view(np.uint8).reshape(-1, 8).astype(np.int16)


@pytest.fixture(scope="module")
def eng(data_dir):
    m = A.VariantModel(f"{data_dir}/simple.alphabet.tsv", A.Weights(), device=-1)
    m.read_lexicon(f"{data_dir}/eng.aspell.lexicon")
    m.build()
    lib = L.lib()
    pv, n = C.POINTER(C.c_uint32)(), C.c_size_t()
    L.check(lib.anx_debug_entries(m.h, C.byref(pv), C.byref(n)))
    ent_vocab = np.ctypeslib.as_array(pv, shape=(n.value,)).copy()
    C.CDLL(None).free(pv)
    sig = np.zeros(n.value, dtype=np.uint64)
    lens = np.zeros(n.value, dtype=np.int32)
    s = C.c_uint64()
    for e, v in enumerate(ent_vocab):
        text = lib.anx_model_vocab_text(m.h, int(v))
        L.check(lib.anx_debug_signature(m.h, text, C.byref(s)))
        sig[e] = s.value
    sb = _sig_bytes(sig)
    lens[:] = sb.sum(axis=1)
    return m, sig, sb, lens


def _lists(m, closure, budget, sigs):
    lib = L.lib()
    sigs = np.asarray(sigs, dtype=np.uint64)
    cum = np.zeros((len(sigs), 8), dtype=np.uint32)
    ids = C.POINTER(C.c_uint32)()
    stats = (C.c_uint64 * 7)()
    L.check(lib.anx_debug_adjacency(m.h, closure, budget, sigs.ctypes.data_as(C.c_void_p), len(sigs), cum.ctypes.data_as(C.c_void_p), C.byref(ids), stats))
    total = 0
    for c in cum:
        if c[0] != 0xFFFFFFFF:
            total = max(total, (int(c[0]) + int(c[7])) * 64)
    arr = np.ctypeslib.as_array(ids, shape=(max(total, 1),)).copy()
    C.CDLL(None).free(ids)
    return cum, arr, list(stats)


def _check_list(sig_u, cum, ids, sb, lens, nent):
    ub = _sig_bytes([sig_u])[0]
    L0 = int(ub.sum())
    dist = np.abs(sb - ub).sum(axis=1)
    want = np.nonzero(dist <= 3)[0]
    row0 = int(cum[0])
    got_all = []
    for s in range(7):
        r0 = row0 + (int(cum[s]) if s else 0)
        r1 = row0 + int(cum[s + 1])
        sec = ids[r0 * 64:r1 * 64]
        real = sec[sec != nent]
        assert (lens[real] == L0 - 3 + s).all()
        assert (np.diff(real.astype(np.int64)) > 0).all()       # ascending entry ids: every entry once
        n = len(real)
        assert (sec[:n] == real).all() and (sec[n:] == nent).all()   # padding only at the end of the section
        assert r1 - r0 == (n + 63) // 64
        got_all.append(real)
    got = np.sort(np.concatenate(got_all))
    assert np.array_equal(got, want)


def test_lists_equal_brute_force(eng):
    m, sig, sb, lens = eng
    rng = np.random.default_rng(5)
    lexsigs = np.unique(sig)
    pick = list(rng.choice(lexsigs, 40, replace=False))
    # neighbours at distance 1 and 2 of lexicon signatures (the closure), and a far-away one
    for s in rng.choice(lexsigs, 20, replace=False):
        b = _sig_bytes([s])[0].copy()
        g = int(rng.integers(0, 7))
        b[g] += 1
        pick.append(int(b.astype(np.uint8).view(np.uint64)[0]))
        h = int(rng.integers(0, 7))
        if b[h] > 0:
            b[h] -= 1
            pick.append(int(b.astype(np.uint8).view(np.uint64)[0]))
    far = int(np.array([40, 40, 40, 40, 40, 40, 40, 0], dtype=np.uint8).view(np.uint64)[0])
    pick.append(far)
    cum, ids, stats = _lists(m, 2, 1 << 40, pick)
    assert stats[0] == len(lexsigs) and stats[2] == stats[1] >= stats[0]
    lexset = set(int(x) for x in lexsigs)
    for i, s in enumerate(pick):
        if int(s) in lexset:
            assert cum[i][0] != 0xFFFFFFFF
        if int(s) == far:
            assert cum[i][0] == 0xFFFFFFFF
        if cum[i][0] != 0xFFFFFFFF:
            _check_list(int(s), cum[i], ids, sb, lens, len(sig))


def test_budget_keeps_the_lexicon_signatures_first(eng):
    m, sig, sb, lens = eng
    lexsigs = np.unique(sig)
    _, _, full = _lists(m, 1, 1 << 40, lexsigs[:4])
    budget = int(full[4]) * 64 * 12 // 3     # a third of what everything needs
    cum, ids, st = _lists(m, 1, budget, lexsigs)
    assert st[4] * 64 * 12 <= budget and 0 < st[2] < st[1]
    kept = [i for i in range(len(lexsigs)) if cum[i][0] != 0xFFFFFFFF]
    assert kept
    for i in kept[:: max(1, len(kept) // 25)]:
        _check_list(int(lexsigs[i]), cum[i], ids, sb, lens, len(sig))
    cum0, _, st0 = _lists(m, 0, 1 << 40, lexsigs[:16])
    assert st0[1] == st0[2] == len(lexsigs) and (cum0[:, 0] != 0xFFFFFFFF).all()
