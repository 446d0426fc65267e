#!/bin/bash
# Experiment (GPU box): rocprofv3 kernel stats of the FM k = 64 + FTRL power-law step
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/fm64s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fm64s -- python $R/bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --steps 12 --warmup 4 --no-cpu-baseline --batches 4 --repeats 2 > /tmp/fm64s.json 2>/tmp/fm64s.err
python3 - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/fm64s/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    m = re.search(r"k_\w+(<[^>]*>)?", r["Name"])
    print("%-46s calls %4s avg %9.1f us  %5s%%" % ((m.group(0) if m else r["Name"][:40])[:46], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
