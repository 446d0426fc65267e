#!/usr/bin/env python3
"""Experiment (GPU box): the config-2 LR step with every second layer of the gradient + Push
kernel's first round of workgroups started late (xf_tune lr_gradient = 4..7: 2..8 naps of
s_sleep 127).  The hypothesis: the first round's workgroups run their two phases in lockstep —
TA-bound loss gathers, then the memory-bound state pass — which is why the phases add up; started
half a lifetime apart they would overlap.  Measured (call 13): 71.5 us without, 71.5 / 75.6 / 76.9 /
77.0 us with 2 / 4 / 6 / 8 naps — no overlap gained, the delay just added.  The switch existed in
the library for that one call only (commit history); this script stays as the record."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xflow_amd import capi  # noqa: E402
from xflow_amd.single import SingleGpuTrainer  # noqa: E402

nkeys = 10_000_000
args = argparse.Namespace(seed=20260926, rows=50000, nnz_per_row=200, batches=8, zipf=0.0,
                          signal_keys=32, keys_per_gpu=nkeys)
keytab = bench.make_key_table(nkeys)
batches = bench.make_batches(args, 0, nkeys, keytab)
tr = SingleGpuTrainer(model="lr", optimizer="ftrl", capacity=2 * nkeys + 1024)
comp = [tr.compile(*b) for b in batches]
for c in comp:
    tr.predict(c)
tr.check()
tr.defrag()
for c in comp:
    tr.predict(c)
for rep in range(2):
    for v in (0, 4, 5, 6, 7, 0):
        capi.tune("lr_gradient", v)
        per, kern = bench._lr_leg_run(tr, comp, steps=40)
        capi.tune("lr_gradient", 0)
        print("lr_gradient=%d  %.4f ms/step (min of 3)  forward %.1f  gradient %.1f us" % (
            v, min(per), kern["forward"] * 1e3, kern["gradient"] * 1e3), flush=True)
