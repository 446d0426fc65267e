#!/bin/bash
# round 6, call 51: kernels of the sequential cycle on a Zipf(1.1) stream
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
SEQ="--force-sharded --general-path --schedule sequential --no-cpu-baseline --steps 4 --warmup 2 --repeats 0 --batches 4 --no-owner-leg --key-build-steps 16 --zipf 1.1 --signal-keys 0"
rm -rf /tmp/_p
(cd /tmp && PYTHONPATH=$GRAFT_REPO_ROOT timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_p -- python $GRAFT_REPO_ROOT/bench.py $SEQ > /tmp/_p.out 2>&1)
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/_p/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:30]:
        print("%-70s calls %4s avg %8.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
