#!/bin/bash
# Survey (GPU box): which kernels of a workload take long for the bytes they move?  One
# rocprofv3 kernel-stats run and two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel trace only) of
# the same command, then every kernel's average time next to its traffic — a kernel far below
# ~3 TB/s with little traffic is waiting on something else (dependent round trips, divergence,
# one long chain): that is how the FM heavy-key kernels were found (217 us for 45 MB).
#   bash tools/survey_kernels.sh gpurun_out/survey "python bench.py --zipf 1.1 --no-cpu-baseline --no-fm-leg --repeats 0 --pmc-calibrate"
set -u
OUT=$1; CMD=$2
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- $CMD > $R/$OUT/stats.json 2> $R/$OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -- $CMD > $R/$OUT/pmc_$c.json 2> $R/$OUT/pmc_$c.err
done
cd $R
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_traffic.json
python3 - $OUT <<'PY'
import csv, glob, json, re, sys
out = sys.argv[1]
tr = json.load(open(out + "/pmc_traffic.json"))["kernels"]
f = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", r["Name"])
    if not m:
        continue
    name = m.group(1) + (m.group(2) or "")
    e = tr.get(name)
    us = float(r["AverageNs"]) / 1e3
    mb = e["traffic"] / 1e6 if e else float("nan")
    rows.append((float(r["TotalDurationNs"]), name, int(r["Calls"]), us, mb,
                 mb / us if us > 0 and e else float("nan")))   # MB per us = TB/s
print("%-46s %6s %10s %10s %8s" % ("kernel", "calls", "avg us", "MB/launch", "TB/s"))
for _, name, calls, us, mb, tbs in sorted(rows, reverse=True)[:24]:
    print("%-46s %6d %10.1f %10.1f %8.2f" % (name[:46], calls, us, mb, tbs))
PY
