#!/usr/bin/env python3
"""Experiment driver (GPU box): the key build of the config-2 shape (xf_batch_compile_local_dev
on raw keys resident in HBM, table settled) under exp_knob variants, timed per call with the
host clock around a stream sync.  Run under rocprofv3 --kernel-trace --stats for per-kernel
times.  Not a benchmark.
  python tools/kb_knobs.py --knobs 0,101,102 [--rows 50000 --nnz-per-row 200 --keys 10000000]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

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
    ap.add_argument("--knobs", default="0")
    ap.add_argument("--zipf", type=float, default=0.0)
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--nnz-per-row", type=int, default=200)
    ap.add_argument("--keys", type=int, default=10_000_000)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--step", action="store_true", help="also run the LR step per minibatch")
    ap.add_argument("--pmc-calibrate", action="store_true",
                    help="after the run, stream known byte counts (for rocprofv3 --pmc passes)")
    a = ap.parse_args()
    import torch
    args = argparse.Namespace(seed=20260926, rows=a.rows, nnz_per_row=a.nnz_per_row,
                              batches=a.batches, zipf=a.zipf)
    keytab = capi.hash_decimal_range(0, a.keys)
    batches = bench.make_batches(args, 0, a.keys, keytab)
    tr = SingleGpuTrainer(model="lr", optimizer="ftrl", capacity=2 * a.keys + 1024)
    comp = [tr.compile(*b) for b in batches]
    tr.check()
    tr.defrag()
    del comp
    L = capi.lib()
    raw = []
    for rowptr, keys, labels in batches:
        raw.append((torch.from_numpy(keys.view(np.int64)).cuda(),
                    torch.from_numpy(rowptr.astype(np.uint32).view(np.int32)).cuda(),
                    torch.from_numpy(labels).cuda(), len(labels), len(keys)))

    def one(i):
        k, rp, lb, R, NNZ = raw[i % len(raw)]
        h = capi.vp()
        capi.check(L.xf_batch_compile_local_dev(C.byref(h), tr.w.h, k.data_ptr(), rp.data_ptr(),
                                                lb.data_ptr(), R, NNZ, 0, None))
        if a.step:
            capi.check(L.xf_lr_step(tr.w.h, h, tr.ws.h, None))
        capi.stream_sync()
        return h

    for knob in [int(x) for x in a.knobs.split(",")]:
        exp_knob(knob)
        for i in range(3):
            L.xf_batch_free(one(i))
        torch.cuda.synchronize()
        ts = []
        for i in range(a.iters):
            t0 = time.perf_counter()
            h = one(i)
            ts.append(time.perf_counter() - t0)
            if i == 0:
                info = (C.c_uint32 * 8)()
                capi.check(L.xf_batch_cells_info(h, info))
            L.xf_batch_free(h)
        ts.sort()
        print(json.dumps({"knob": knob, "median_us": round(ts[len(ts) // 2] * 1e6, 1),
                          "min_us": round(ts[0] * 1e6, 1), "segments": info[5],
                          "nitems": info[4]}), flush=True)
    exp_knob(0)
    tr.check()
    if a.pmc_calibrate:
        for kind in range(10):
            capi.check(L.xf_calib_stream(kind, 1 << 30, 3))
    print(json.dumps({"config": {"workload": "key build: LR+FTRL, %d keys settled, %d rows x %d "
                                 "nnz per minibatch%s" % (a.keys, a.rows, a.nnz_per_row,
                                                          ", zipf %.2f" % a.zipf if a.zipf else
                                                          ", uniform")}}), flush=True)


if __name__ == "__main__":
    main()
