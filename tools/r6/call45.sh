#!/bin/bash
# round 6, call 45: where xf_sort_key_pos spends its time on a Zipf(1.1) stream (kernel stats)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
cat > /tmp/zs.py <<'PY'
import sys
import numpy as np
from xflow_amd import capi
rng = np.random.RandomState(1)
n = 10_000_000
pool = rng.randint(0, 2**63, size=n).astype(np.uint64) * np.uint64(2) + np.uint64(1)
if sys.argv[1] == "zipf":
    keys = pool[np.minimum(rng.zipf(1.1, size=n), n) - 1]
else:
    keys = pool[rng.randint(0, int(n * 0.8), size=n)]
    hk = rng.randint(0, 2**63, size=32).astype(np.uint64) * np.uint64(2)
    keys[rng.randint(0, n, size=50000)] = hk[rng.randint(0, 32, size=50000)]
print(capi.sort_key_pos(keys, repeat=10)[2:])
PY
for w in zipf hot; do
rm -rf /tmp/_p
(cd /tmp && PYTHONPATH=$GRAFT_REPO_ROOT timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_p -- python /tmp/zs.py $w 2>&1 | tail -2)
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/_p/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
