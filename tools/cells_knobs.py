#!/usr/bin/env python3
"""Experiment driver (GPU box): the LR step of the config-2 shape under the exp_knob variants
of the cells kernels (parts switched off to see what each part costs).  Not a benchmark.
The timing-only variants (knobs 308 ... 400: they give WRONG tables) exist only in a library built
with XF_EXTRA_FLAGS=-DXF_EXPERIMENTS (python -m xflow_amd.build, or tools/build_variant.sh); the
product library refuses them."""
import argparse
import json
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
    ap.add_argument("--knobs", default="0")
    ap.add_argument("--zipf", type=float, default=0.0)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--signal-keys", type=int, default=0)
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
    print(json.dumps(comp[0].cells_info()))
    for knob in [int(x) for x in a.knobs.split(",")]:
        exp_knob(knob)
        for i in range(4):
            tr.step(comp[i % len(comp)])
        tr.check()
        tr.profile(True)
        for i in range(a.steps):
            tr.step(comp[i % len(comp)])
        ms, n = tr.profile_read()
        tr.profile(False)
        print("knob %3d  forward %.1f us  gradient %.1f us" % (
            knob, ms["forward"] / n * 1e3, ms["gradient"] / n * 1e3), flush=True)
    exp_knob(0)


if __name__ == "__main__":
    main()
