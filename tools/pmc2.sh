#!/bin/bash
# Two rocprofv3 --pmc passes (L2 -> fabric read / write requests by size, kernel-trace only) of a
# command that ends with the known-traffic calibration kernels, and the per-kernel traffic:
#   bash tools/pmc2.sh <outdir> <name> <command ... --pmc-calibrate>
# -> <outdir>/pmc_traffic_<name>.json (+ the two counter_collection.csv files)
set -u
OUT=$1; NAME=$2; shift 2
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace \
    --output-format csv -d $R/$OUT/${NAME}_rd -- "$@" > $R/$OUT/${NAME}_rd.json 2> $R/$OUT/${NAME}_rd.err
echo "pmc2 $NAME rd rc=$?"
timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace \
    --output-format csv -d $R/$OUT/${NAME}_wr -- "$@" > $R/$OUT/${NAME}_wr.json 2> $R/$OUT/${NAME}_wr.err
echo "pmc2 $NAME wr rc=$?"
cd $R
python tools/pmc_traffic2.py $OUT/${NAME}_rd $OUT/${NAME}_wr > $OUT/pmc_traffic_$NAME.json
cp $(find $OUT/${NAME}_rd -name "*counter_collection.csv" | head -1) $OUT/pmc_rd_${NAME}_counter_collection.csv 2>/dev/null
cp $(find $OUT/${NAME}_wr -name "*counter_collection.csv" | head -1) $OUT/pmc_wr_${NAME}_counter_collection.csv 2>/dev/null
rm -rf $OUT/${NAME}_rd $OUT/${NAME}_wr
python - <<PY
import json
d=json.load(open("$OUT/pmc_traffic_$NAME.json"))
print(d["workload"])
for k,e in d["calibration"].items():
    print("  calib %-34s measured/useful %6.3f  measured/lines %6.3f  %7.1f us  measured %6.0f GB/s" % (k, e["measured_over_useful"], e["measured_over_lines"], e["median_us"], e["measured_gbs"]))
for k,e in d["kernels"].items():
    if e["traffic"] > 1e6:
        print("  %-44s read %8.1f MB write %8.1f MB  %7.1f us  %6.0f GB/s  (32B %.2f, 128B %.2f of the read requests)" % (k[:44], e["read"]/1e6, e["write"]/1e6, e["median_us_under_pmc"], e["gbs_under_pmc"], e["read_requests"]["share_32B"], e["read_requests"]["share_128B"]))
PY
