#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, two passes, in-run stream calibration) of the FM step's
# kernels:   bash tools/pmc_fm_traffic.sh gpurun_out/pmc_fm_traffic "--k 16 --optimizer sgd"
# (FMLEG=1: bench.py's `fm` leg alone — tools/fm_leg.py, k = 16 + SGD — instead of bench.py --model fm)
set -u
OUT=$1; ARGS=${2:---k 16 --optimizer sgd}
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  if [ -n "${FMLEG:-}" ]; then CMD="python $R/tools/fm_leg.py --batches 4 --pmc-calibrate"
  else CMD="python $R/bench.py --model fm $ARGS --steps 6 --warmup 4 --no-cpu-baseline --pmc-calibrate"; fi
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -- \
      $CMD > $R/$OUT/pmc_$c.json 2> $R/$OUT/pmc_$c.err
done
cd $R
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_traffic.json
python - <<PY
import json
d=json.load(open("$OUT/pmc_traffic.json"))
for k,e in d["kernels"].items():
    if "fm_" in k or "gather" in k or "resolve" in k or "pull" in k:
        print("%-40s launches %3d  fetch %8.1f MB  write %8.1f MB  traffic %8.1f MB  %7.1f us" % (k[:40], e["launches"], e["fetch_corrected"]/1e6, e["write_corrected"]/1e6, e["traffic"]/1e6, e["median_us_under_pmc"]))
PY
