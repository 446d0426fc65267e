#!/usr/bin/env python3
"""Experiment driver (GPU box): bench.py's FM leg alone (FM k = 16 + SGD on the config-2 row
shape, BASELINE configs[3]) — what `python bench.py` reports as its `fm` object, without the LR
run and the CPU baseline around it.  XF_FM_TABLE_RECORDS=0 in the environment times the step
with the per-step factor gather instead of the table-resident records.
  python tools/fm_leg.py [bench.py's flags: --rows --nnz-per-row --keys-per-gpu --zipf ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    args = bench.parse_args()
    if not args.keys_per_gpu:
        args.keys_per_gpu = 10_000_000
    from xflow_amd import capi
    capi.require_gpu()
    keytab = bench.make_key_table(args.keys_per_gpu)
    batches = bench.make_batches(args, 0, args.keys_per_gpu, keytab)
    # FM_KNOBS=0,301,...: the leg once per exp_knob value (experiments: a library built with
    # XF_EXTRA_FLAGS=-DXF_EXPERIMENTS; 0 = the product's kernels, any library)
    for knob in [int(x) for x in os.environ.get("FM_KNOBS", "0").split(",")]:
        if knob:
            capi.tune("exp_knob", knob)
        out = bench.fm_leg(args, batches)
        out["exp_knob"] = knob
        print(json.dumps(out), flush=True)
    if args.pmc_calibrate:   # known byte counts for the rocprofv3 --pmc passes
        import ctypes  # noqa: F401
        for kind in range(10):
            capi.check(capi.lib().xf_calib_stream(kind, 1 << 30, 3))
        print(json.dumps({"config": {"workload": out["workload"]}}), flush=True)


if __name__ == "__main__":
    main()
