#!/usr/bin/env python3
"""Per-kernel timings of the hot path at the BASELINE config-2 shape, each kernel launched
on its own through the device-pointer C ABI and timed with HIP events on the same stream.
Tuning aid (not the bench): python tools/microbench.py [--load-factor 0.5] [--reps 10]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xflow_amd import capi  # noqa: E402
from xflow_amd.sharded import HipStages  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--nnz", type=int, default=200)
    ap.add_argument("--keys", type=int, default=10_000_000)
    ap.add_argument("--load-factor", type=float, default=0.5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--keybuild", action="store_true")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    st = HipStages("lr", "ftrl", 0, int(a.keys / a.load_factor) + 1024, 0, 1)
    keytab = capi.hash_decimal_range(0, a.keys)
    rng = np.random.RandomState(1)
    batches = []
    for _ in range(a.batches):
        fid = rng.randint(0, a.keys, size=a.rows * a.nnz)
        rowptr = np.arange(a.rows + 1, dtype=np.uint64) * np.uint64(a.nnz)
        batches.append(st.compile_batch(rowptr, keytab[fid],
                                        rng.randint(0, 2, size=a.rows).astype(np.int32)))
    tw = st.w
    s = torch.cuda.current_stream().cuda_stream
    L = capi.lib()

    def timed(name, fn, bytes_):
        for b in batches:          # warm (also inserts keys)
            fn(b)
        torch.cuda.synchronize()
        ts = []
        for r in range(a.reps):
            b = batches[r % len(batches)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(b)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        us = float(np.median(ts))
        print("%-28s %8.1f us   %7.1f MB alg  %6.2f TB/s" % (name, us, bytes_ / 1e6,
                                                           bytes_ / us / 1e6))

    b0 = batches[0]
    U, NNZ, R = b0.U, b0.NNZ, b0.R
    rows = st.empty(U, torch.int32)
    wu = st.empty(U, torch.float32)
    loss = st.empty(R, torch.float32)
    g = st.empty(U, torch.float32)
    timed("resolve", lambda b: tw.resolve_dev(b.ukeys.data_ptr(), b.U, rows.data_ptr(), s), 20 * U)
    timed("pull (resolve+gather)", lambda b: capi.check(L.xf_table_pull_dev(
        tw.h, b.ukeys.data_ptr(), b.U, rows.data_ptr(), wu.data_ptr(), s)), 28 * U)
    timed("gather", lambda b: tw.gather_dev(rows.data_ptr(), b.U, wu.data_ptr(), s), 12 * U)
    timed("forward", lambda b: capi.check(L.xf_lr_forward_dev(
        capi.C.byref(b.view), wu.data_ptr(), loss.data_ptr(), None, s)), 8 * NNZ + 12 * R)
    timed("gradient (tiled)", lambda b: capi.check(L.xf_lr_grad_dev(
        capi.C.byref(b.view), loss.data_ptr(), g.data_ptr(), s)), 8 * NNZ + 8 * U)

    def upd(b):
        tw.resolve_dev(b.ukeys.data_ptr(), b.U, rows.data_ptr(), s)
    timed("update (rows of batch 0)", lambda b: tw.update_dev(rows.data_ptr(), U, g.data_ptr(), s),
          32 * U)

    def gu(b):
        capi.check(L.xf_table_pull_dev(tw.h, b.ukeys.data_ptr(), b.U, rows.data_ptr(),
                                       wu.data_ptr(), s))
        capi.check(L.xf_lr_grad_update_dev(tw.h, capi.C.byref(b.view), rows.data_ptr(),
                                           wu.data_ptr(), loss.data_ptr(), g.data_ptr(), s))
    timed("pull + grad_update", gu, 28 * U + 8 * NNZ + 36 * U)
    st.check()


if __name__ == "__main__" and "--keybuild" not in sys.argv:
    main()


def keybuild():
    """device key build (a3) at the config-2 shape: raw CSR already in HBM"""
    import ctypes as C
    import time
    torch.cuda.set_device(0)
    R, nnz, K = 50000, 200, 10_000_000
    keytab = capi.hash_decimal_range(0, K)
    rng = np.random.RandomState(2)
    keys = torch.from_numpy(keytab[rng.randint(0, K, size=R * nnz)].view(np.int64)).cuda()
    rowptr = torch.arange(R + 1, dtype=torch.int32, device="cuda") * nnz
    labels = torch.zeros(R, dtype=torch.int32, device="cuda")
    L = capi.lib()
    if "--no-pool" in sys.argv:
        capi.tune("batch_pool_blobs", 0)
    ts = []
    for it in range(6):
        h = capi.vp()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        capi.check(L.xf_batch_compile_dev(C.byref(h), keys.data_ptr(), rowptr.data_ptr(),
                                          labels.data_ptr(), R, R * nnz, None))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        L.xf_batch_free(h)
    print("device key build (sort+unique+views+tiles+panels): %.2f ms  (min %.2f)" % (
        float(np.median(ts[1:])) * 1e3, min(ts[1:]) * 1e3))
    t0 = time.perf_counter()
    hb = capi.Batch(np.arange(R + 1, dtype=np.uint64) * np.uint64(nnz),
                    keytab[rng.randint(0, K, size=R * nnz)], np.zeros(R, np.int32))
    print("host key build (%d threads): %.0f ms" % (os.cpu_count(), (time.perf_counter() - t0) * 1e3))


if __name__ == "__main__" and "--keybuild" in sys.argv:
    keybuild()
