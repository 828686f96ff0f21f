#!/usr/bin/env python3
"""A few small calls of one size under rocprofv3 --kernel-trace: small_trace.py N [reps]"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.fresh_batch import setup  # noqa: E402


def main():
    n = int(sys.argv[1])
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    torch, A, model, queries, params, paths = setup(nq=max(n, 1000))
    from analiticcl_amd import _lib as LL
    L = A.lib()
    enc = [q.encode("utf-8") for q in queries[:n]]
    arr = (C.c_char_p * n)(*enc)
    cp = params._c()
    ts = []
    for _ in range(reps):
        rows = C.POINTER(LL.Result)()
        offs = C.POINTER(C.c_size_t)()
        t = time.perf_counter()
        assert L.anx_find_variants_batch(model.h, arr, n, C.byref(cp), C.byref(rows), C.byref(offs)) == 0
        ts.append(time.perf_counter() - t)
        L.anx_results_free(rows, offs)
    print("n", n, "best us", min(ts) * 1e6, "median", sorted(ts)[len(ts) // 2] * 1e6)


if __name__ == "__main__":
    main()
