#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE passes (separate runs, --pmc with --kernel-trace only) of the
# key-build loop with the in-run stream calibration; per-kernel traffic -> $1/pmc_traffic_key_build.json
set -u
OUT=$1
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/kbpmc_$c -- \
      python $R/tools/kb_knobs.py --knobs 0 --iters 8 --step --pmc-calibrate \
      > $R/$OUT/kbpmc_$c.json 2> $R/$OUT/kbpmc_$c.err
done
cd $R
python tools/pmc_traffic.py $OUT/kbpmc_FETCH_SIZE $OUT/kbpmc_WRITE_SIZE > $OUT/pmc_traffic_key_build.json
cp $(find $OUT/kbpmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $OUT/pmc_fetch_key_build_counter_collection.csv
cp $(find $OUT/kbpmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/pmc_write_key_build_counter_collection.csv
rm -rf $OUT/kbpmc_FETCH_SIZE $OUT/kbpmc_WRITE_SIZE
python - <<PY
import json
d=json.load(open("$OUT/pmc_traffic_key_build.json"))
print(d["calibration_true_bytes_per_counted_byte"])
for k,e in d["kernels"].items():
    if "k_kb" in k or "k_lr" in k:
        print("%-40s fetch %7.1f MB  write %7.1f MB  traffic %7.1f MB  %6.1f us" % (k[:40], e["fetch_corrected"]/1e6, e["write_corrected"]/1e6, e["traffic"]/1e6, e["median_us_under_pmc"]))
PY
