#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/pmc_FETCH_SIZE -- \
      python bench.py --steps 6 --warmup 8 --no-cpu-baseline --pmc-calibrate
  rocprofv3 --pmc WRITE_SIZE ... (same command)
  python tools/pmc_traffic.py out/pmc_FETCH_SIZE out/pmc_WRITE_SIZE > profiles/rNN/pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md §HBM, §rocprofv3 PMC slots): the counters are in
KiB; FETCH_SIZE and WRITE_SIZE need separate passes (3 + 2 of the 4 TCC slots); on gfx950
FETCH_SIZE reads 1/2 of a wide coalesced stream and other widths are uncalibrated, so every
pass also runs xf_calib_stream (1 GiB streams, beyond the 256 MiB Infinity Cache, read/write
at 4/8/16 B per lane) and the factor true_bytes / (counter * 1024) measured there for the
access width a kernel uses is applied to that kernel's counter.
"""
import csv
import glob
import json
import os
import re
import sys

CALIB_BYTES = 1 << 30
# access width (bytes per lane) that dominates each kernel's reads / writes
WIDTH = {
    "k_resolve": (8, 4), "k_pull_settled": (8, 4), "k_gather": (4, 4), "k_update": (4, 4), "k_lr_forward": (4, 4),
    "k_lr_forward_panel": (4, 8), "k_lr_forward_tiled": (4, 8), "k_lr_finalize": (8, 4),
    "k_lr_grad": (4, 4), "k_lr_grad_tiled": (4, 4),
    "k_lr_grad_update": (4, 4), "k_lr_grad_heavy": (4, 4), "k_fm_forward": (4, 4),
    "k_fm_grad": (4, 4),
    # round 2: cells kernels — forward reads 4-byte entries / weights and writes fp64 partials,
    # finalize reads fp64 partials, gradient reads 4-byte entries / losses / w and 8-byte {n,z}
    "k_lr_fwd_cells": (4, 8), "k_lr_finalize_cells": (8, 4), "k_lr_grad_cells": (4, 4),
    "k_lr_grad_split_finish": (8, 4), "k_cell_keys": (4, 4),
    # round 3: the keyed build — histogram reads 8-byte keys; scatter reads keys 16 bytes per
    # lane and writes 12-byte records (priced with the 16-byte calibration); resolve reads
    # 12-byte records and the table's keys 16 bytes per lane and writes 4-byte entries
    "k_kb_hist": (8, 4), "k_kb_scatter": (16, 16), "k_kb_resolve": (16, 4), "k_kb_scan": (4, 4),
    # FM: factor rows / records move as 16-byte accesses
    "k_fm_gather_scalars": (16, 16), "k_fm_forward_scalars": (16, 4), "k_fm_grad_tiled": (16, 16),
    "k_fm_row_partials": (16, 8), "k_fm_ridx": (4, 4),
}


def kname(full):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", full)
    if not m:
        return full.split("(")[0].strip()
    return m.group(1) + (m.group(2) or "")


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = {}
    for r in csv.DictReader(open(f)):
        rows.setdefault(kname(r["Kernel_Name"]), []).append(
            (float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    return rows


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    calib = {}
    for w, t in ((4, "unsigned int"), (8, "unsigned long"), (16, "HIP_vector_type<unsigned int, 4u>")):
        kr = [k for k in fetch if k.startswith("k_calib_read") and t in k]
        kw = [k for k in write if k.startswith("k_calib_write") and t in k]
        if kr:
            calib["read%d" % w] = CALIB_BYTES / (median([x[0] for x in fetch[kr[0]]]) * 1024)
        if kw:
            calib["write%d" % w] = CALIB_BYTES / (median([x[0] for x in write[kw[0]]]) * 1024)
    workload = None
    for d in sys.argv[1:3]:   # the bench's own JSON line, written next to the pass directory
        try:
            workload = json.loads(open(d.rstrip("/") + ".json").read().strip().splitlines()[-1])[
                "config"]["workload"]
            break
        except (OSError, ValueError, KeyError, IndexError):
            pass
    out = {"workload": workload,
           "unit": "bytes per launch (median over steady-state launches)",
           "calibration_true_bytes_per_counted_byte": calib, "kernels": {}}
    for k in sorted(set(fetch) & set(write)):
        if k.startswith("k_calib") or not k.startswith("k_"):
            continue
        base = k.split("<")[0]
        rw, ww = WIDTH.get(base, (4, 4))
        # steady state = the later half of the launches (the first ones insert keys)
        fv = [x[0] for x in fetch[k]][len(fetch[k]) // 2:]
        wv = [x[0] for x in write[k]][len(write[k]) // 2:]
        fraw, wraw = median(fv) * 1024, median(wv) * 1024
        fcor = fraw * calib.get("read%d" % rw, 1.0)
        wcor = wraw * calib.get("write%d" % ww, 1.0)
        out["kernels"][k] = {"launches": len(fetch[k]), "fetch_raw": fraw, "write_raw": wraw,
                             "fetch_corrected": fcor, "write_corrected": wcor,
                             "traffic": fcor + wcor,
                             "median_us_under_pmc": median([x[1] for x in fetch[k]])}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
