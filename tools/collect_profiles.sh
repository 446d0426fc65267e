#!/bin/bash
# gpurun_out/<run> (written by tools/profile_round<N>.sh on the GPU box) -> profiles/r<N> (tracked):
# the JSON lines and summaries as they are, the raw counter files cut down to the kernels the
# numbers are quoted for (the step's, the key build's and the calibration kernels).
#   bash tools/collect_profiles.sh gpurun_out/r05 profiles/r05
set -e
SRC=$1; DST=$2
mkdir -p $DST
cp $SRC/bench_*.json $SRC/pmc_traffic_*.json $SRC/*_kernel_stats.csv $SRC/fm_leg.json $SRC/e2e.json $DST/ 2>/dev/null || true
for f in $SRC/pmc_rd_*_counter_collection.csv $SRC/pmc_wr_*_counter_collection.csv; do
  [ -f "$f" ] || continue
  python3 - "$f" "$DST/$(basename $f)" <<'PY'
import csv, sys
keep = ("k_lr_", "k_fm_", "k_kb_", "k_eb_", "k_ar_", "k_calib", "k_owner", "k_sum_partials")
rows = list(csv.reader(open(sys.argv[1])))
hdr, body = rows[0], rows[1:]
ki = hdr.index("Kernel_Name")
seen = {}
out = [hdr]
for r in body:
    k = r[ki]
    if not any(x in k for x in keep):
        continue
    key = (k, r[hdr.index("Counter_Name")])
    seen[key] = seen.get(key, 0) + 1
    if seen[key] <= 12:          # a dozen launches per (kernel, counter) are plenty
        out.append(r)
csv.writer(open(sys.argv[2], "w")).writerows(out)
PY
done
cp $DST/pmc_traffic_lr.json profiles/pmc_traffic_latest.json
ls -la $DST | awk '{s+=$5} END {print "profiles bytes:", s}'
