#!/usr/bin/env python3
"""Experiment driver (GPU box): the phases of the general gradient kernel's work items
(k_lr_grad_cells, -DXF_GRAD_TIMELINE build: `XF_EXTRA_FLAGS=-DXF_GRAD_TIMELINE python -m
xflow_amd.build` before, a plain rebuild after).  Not a benchmark."""
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
    ap.add_argument("--zipf", type=float, default=1.1)
    ap.add_argument("--batches", type=int, default=16)
    ap.add_argument("--signal-keys", type=int, default=32)
    ap.add_argument("--knob", type=int, default=0)
    a = ap.parse_args()
    args = argparse.Namespace(seed=20260926, rows=50000, nnz_per_row=200, batches=a.batches,
                              zipf=a.zipf, signal_keys=a.signal_keys, keys_per_gpu=10_000_000)
    nkeys = 10_000_000
    keytab = capi.hash_decimal_range(0, nkeys)
    batches = bench.make_batches(args, 0, nkeys, keytab)
    tr = SingleGpuTrainer(model="lr", optimizer="ftrl", capacity=2 * nkeys + 1024)
    comp = [tr.compile(*b) for b in batches]
    tr.check()
    tr.defrag()
    for c in comp:
        tr.predict(c)
    info = comp[0].cells_info()
    print(info)
    exp_knob(a.knob)
    for i in range(6):
        tr.step(comp[i % len(comp)])
    tr.step(comp[0])
    tr.check()
    n = min(info["nitems"], 16384)
    buf = np.zeros(n * 8, np.uint64)
    L = capi.lib()
    L.xf_debug_grad_timeline.argtypes = [C.c_void_p, C.c_size_t]
    assert L.xf_debug_grad_timeline(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(n, 8).astype(np.int64)
    t0 = t[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    total = t[:, 7]
    end = np.maximum(np.maximum(t[:, 5], t[:, 6]), t[:, 3])
    print("kernel span %.1f us; first start %.1f last start %.1f" % (
        us(end.max()), us(t[:, 0].min()), us(t[:, 0].max())))
    split = t[:, 4] == 0
    sparse = t[:, 5] != 0
    for name, m in (("slices / dense-sweep items (no list)", ~sparse), ("listed items", sparse)):
        if not m.any():
            continue
        e = end[m]
        print("%s: %d items, entries avg %.0f; start..cum %.2f  accumulate %.2f  barrier %.2f  "
              "rest %.2f  whole %.2f us (avg)" % (
                  name, m.sum(), total[m].mean(), ((t[m, 1] - t[m, 0]) / 100).mean(),
                  ((t[m, 2] - t[m, 1]) / 100).mean(), ((t[m, 3] - t[m, 2]) / 100).mean(),
                  ((e - t[m, 3]) / 100).mean(), ((e - t[m, 0]) / 100).mean()))
    # occupancy over time: items in flight per 2 us
    edges = np.arange(0, us(end.max()) + 2, 2.0)
    infl = [(np.sum((us(t[:, 0]) <= x) & (us(end) > x))) for x in edges]
    print("in flight per 2 us:", infl)
    big = total > 4096
    d = (end - t[:, 0]) / 100.0
    print("item time percentiles 10/50/90/99/max: %s" % np.percentile(d, [10, 50, 90, 99, 100]).round(1))
    o = np.argsort(t[:, 0])
    q = len(o) // 4
    for i in range(4):
        m = o[i * q:(i + 1) * q]
        print("quarter %d of the starts: start %.1f..%.1f  cum %.2f acc %.2f rest %.2f whole %.2f" % (
            i, us(t[m, 0]).min(), us(t[m, 0]).max(), ((t[m, 1] - t[m, 0]) / 100).mean(),
            ((t[m, 2] - t[m, 1]) / 100).mean(), ((end[m] - t[m, 3]) / 100).mean(), d[m].mean()))
    print("items with > 4096 entries: %d; their whole time avg %.1f us, start avg %.1f us" % (
        big.sum(), ((end[big] - t[big, 0]) / 100).mean() if big.any() else 0,
        us(t[big, 0]).mean() if big.any() else 0))


if __name__ == "__main__":
    main()
