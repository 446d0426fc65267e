#!/bin/bash
# Experiment (GPU box): rocprofv3 kernel stats of the N = 8 owner-side shape through the
# owner-compute exchange path at world 1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/n8s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/n8s -- python $R/bench.py --rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --force-sharded --general-path --schedule owner --no-cpu-baseline --repeats 1 --batches 4 --key-build-steps 0 ${EXTRA:-} > /tmp/n8s.json 2>/tmp/n8s.err
python3 - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/n8s/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    m = re.search(r"k_\w+(<[^>]*>)?", r["Name"])
    print("%-46s calls %4s avg %9.1f us  %5s%%" % ((m.group(0) if m else r["Name"][:40])[:46], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 /tmp/n8s.json | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['kernels_ms'].items() if v})"
