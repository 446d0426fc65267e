#!/usr/bin/env python3
"""Experiment driver (GPU box): the FM update() of a fresh minibatch — xf_batch_compile_dev on
raw keys resident in HBM + xf_fm_step — in a loop, for rocprofv3 --kernel-trace --stats.
  python tools/fm_kb_prof.py [--k 16 --optimizer sgd --iters 12]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from xflow_amd import capi  # noqa: E402
from xflow_amd.single import SingleGpuTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--sorted", action="store_true", help="xf_batch_compile_dev (the sort-based "
                    "build) instead of xf_batch_compile_fm_dev")
    ap.add_argument("--key-build", type=int, default=0, help="xf_tune key_build (1: the "
                    "library's sorts, the tests' second implementation)")
    a = ap.parse_args()
    import torch
    args = argparse.Namespace(seed=20260926, rows=50000, nnz_per_row=200, batches=4, zipf=0.0)
    keys = 10_000_000
    keytab = capi.hash_decimal_range(0, keys)
    batches = bench.make_batches(args, 0, keys, keytab)
    tr = SingleGpuTrainer(model="fm", optimizer=a.optimizer, k=a.k, capacity=2 * keys + 1024)
    comp = [tr.compile(*b) for b in batches]
    for c in comp:
        tr.predict(c)
    tr.check()
    tr.defrag()
    del comp
    capi.tune("min_panel_nnz", 1e18)
    capi.tune("key_build", a.key_build)
    L = capi.lib()
    raw = [(torch.from_numpy(k.view(np.int64)).cuda(),
            torch.from_numpy(rp.astype(np.uint32).view(np.int32)).cuda(),
            torch.from_numpy(lb).cuda(), len(lb), len(k)) for rp, k, lb in batches]
    prev, nk = [None], [0]

    def one(i, step=True):
        k, rp, lb, R, N = raw[i % len(raw)]
        h = capi.vp()
        if a.sorted:
            capi.check(L.xf_batch_compile_dev(C.byref(h), k.data_ptr(), rp.data_ptr(),
                                              lb.data_ptr(), R, N, None))
        else:
            keyed = C.c_int(0)
            capi.check(L.xf_batch_compile_fm_dev(C.byref(h), tr.w.h, tr.v.h, k.data_ptr(),
                                                 rp.data_ptr(), lb.data_ptr(), R, N, None,
                                                 C.byref(keyed)))
            nk[0] += keyed.value
        if prev[0] is not None:
            L.xf_batch_free(prev[0])
        if step:
            capi.check(L.xf_fm_step(tr.w.h, tr.v.h, h, tr.ws.h, None))
        prev[0] = h
    for step in (False, True):
        for i in range(2):
            one(i, step)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.iters):
            one(i, step)
        torch.cuda.synchronize()
        print("%s: %.3f ms per minibatch" % ("compile + step" if step else "compile alone",
                                             (time.perf_counter() - t0) / a.iters * 1e3))
    tr.check()
    print("range-partitioned builds: %d" % nk[0])


if __name__ == "__main__":
    main()
