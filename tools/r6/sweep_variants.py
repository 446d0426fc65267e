#!/usr/bin/env python3
"""Experiment (GPU box): the LR step on a table that holds every key of a large key space, under
the named variants of the gradient + Push kernel (xf_tune lr_gradient / old_weight).  (The run
DESIGN 3 quotes — lr_gradient 4 / 5 / 6 = the dense kernel with its state rows prefetched, with 512
threads per chunk, with both — was made at commit 615221c's successor, where those variants were
product instantiations for the length of the experiment; they live behind -DXF_EXPERIMENTS now:
exp_knob 302 / 304 / 306; likewise the touched keys compacted per wavefront with the idle
slots issuing no loads, exp_knob 301: 310.7 against 309.4 us at 1e8 keys, call 9.)
    python tools/r6/sweep_variants.py [nkeys] [name=value ...]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from xflow_amd import capi  # noqa: E402
from xflow_amd.single import SingleGpuTrainer  # noqa: E402

nkeys = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
variants = [a for a in sys.argv[2:]] or ["lr_gradient=0", "lr_gradient=2", "lr_gradient=3",
                                         "lr_gradient=1"]
args = argparse.Namespace(seed=20260926, rows=50000, nnz_per_row=200, batches=4, zipf=0.0,
                          signal_keys=0, keys_per_gpu=nkeys)
keytab = bench.make_key_table(nkeys)
batches = bench.make_batches(args, 0, nkeys, keytab)
tr = SingleGpuTrainer(model="lr", optimizer="ftrl", capacity=int(nkeys / 0.5) + 1024)
for lo in range(0, nkeys, 10_000_000):
    kk = keytab[lo:lo + 10_000_000]
    rows = max(1, len(kk) // 200)
    rp = np.minimum(np.arange(rows + 1, dtype=np.uint64) * np.uint64(200), np.uint64(len(kk)))
    rp[-1] = len(kk)
    capi.LocalBatch(tr.w, rp, kk, np.zeros(rows, np.int32), retain_keys=False)
tr.check()
t0 = time.perf_counter()
tr.defrag()
print("defrag of %d keys: %.1f ms" % (len(tr.w), (time.perf_counter() - t0) * 1e3), flush=True)
comp = [tr.compile(*b) for b in batches]
for c in comp:
    tr.predict(c)
tr.check()
out = {}
for v in variants:
    name, _, val = v.partition("=")
    capi.tune(name, float(val))
    per, kern = bench._lr_leg_run(tr, comp, steps=16)
    capi.tune(name, 0)
    out[v] = {"ms_per_step": min(per), "kernels_ms": kern}
    print(v, "%.4f ms/step" % min(per), {k: round(x, 4) for k, x in kern.items() if x}, flush=True)
wkb = bench._update_dev_ms(tr, batches)
out["with_key_build_ms"] = wkb
print("with_key_build", wkb)
print(json.dumps(out))
