#!/usr/bin/env python3
"""Experiment driver (GPU box): in-kernel phase timeline of the keyed build (exp_knob = 200:
thread 0 of every workgroup stamps wall_clock64 at the phase boundaries).  Prints, per kernel,
when the workgroups started / ended relative to the kernel's first stamp and the median
duration of every phase.  Not a benchmark.
  python tools/kb_timeline.py [--rows 50000 --nnz-per-row 200 --keys 10000000]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from xflow_amd import capi  # noqa: E402


def exp_knob(v):
    """the experiments' numeric knob: only a library built with XF_EXTRA_FLAGS=-DXF_EXPERIMENTS
    has it (xf_common.h); 0 = the product's choice needs none"""
    try:
        capi.tune("exp_knob", v)
    except capi.XFError:
        if v:
            raise SystemExit("this experiment needs a library built with "
                             "XF_EXTRA_FLAGS=-DXF_EXPERIMENTS python -m xflow_amd.build --force")

from xflow_amd.single import SingleGpuTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--zipf", type=float, default=0.0)
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--nnz-per-row", type=int, default=200)
    ap.add_argument("--keys", type=int, default=10_000_000)
    a = ap.parse_args()
    import torch
    args = argparse.Namespace(seed=20260926, rows=a.rows, nnz_per_row=a.nnz_per_row, batches=2,
                              zipf=a.zipf)
    keytab = capi.hash_decimal_range(0, a.keys)
    batches = bench.make_batches(args, 0, a.keys, keytab)
    tr = SingleGpuTrainer(model="lr", optimizer="ftrl", capacity=2 * a.keys + 1024)
    comp = [tr.compile(*b) for b in batches]
    tr.check()
    tr.defrag()
    del comp
    L = capi.lib()
    rowptr, keys, labels = batches[0]
    k = torch.from_numpy(keys.view(np.int64)).cuda()
    rp = torch.from_numpy(rowptr.astype(np.uint32).view(np.int32)).cuda()
    lb = torch.from_numpy(labels).cuda()

    def one():
        h = capi.vp()
        capi.check(L.xf_batch_compile_local_dev(C.byref(h), tr.w.h, k.data_ptr(), rp.data_ptr(),
                                                lb.data_ptr(), len(labels), len(keys), 0, None))
        capi.stream_sync()
        L.xf_batch_free(h)

    for _ in range(3):
        one()
    exp_knob(200)
    one()
    exp_knob(0)
    cap = 1 << 22
    buf = (C.c_ulonglong * cap)()
    shape = (C.c_uint32 * 3)()
    slots = L.xf_kb_debug_read(buf, cap, shape)
    arr = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    off = 0
    for name, nwg in zip(("hist", "scatter", "resolve"), shape):
        t = arr[off:off + nwg * slots].reshape(nwg, slots)
        off += nwg * slots
        live = t[:, 0] > 0
        t = t[live]
        if not len(t):
            continue
        t0 = t[:, 0].min()
        end = np.where(t > 0, t, 0).max(axis=1)
        print("%s: %d workgroups; start %.1f..%.1f us, end %.1f..%.1f us (median %.1f)" % (
            name, len(t), (t[:, 0].min() - t0) / 100, (t[:, 0].max() - t0) / 100,
            (end.min() - t0) / 100, (end.max() - t0) / 100, (np.median(end) - t0) / 100))
        prev = 0
        for s in range(1, slots):
            ok = (t[:, s] > 0) & (t[:, prev] > 0)
            if not ok.any():
                continue
            d = (t[ok, s] - t[ok, prev]) / 100.0
            print("   slot %2d -> %2d: median %6.2f us  p90 %6.2f  max %6.2f  (n=%d)" % (
                prev, s, np.median(d), np.percentile(d, 90), d.max(), ok.sum()))
            prev = s
    tr.check()


if __name__ == "__main__":
    main()
