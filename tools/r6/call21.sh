#!/bin/bash
# round 6, call 21: the first cells' allocation set aside AFTER the warm-up (which took it)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
for k in 10000000 100000000; do
  timeout 600 python tools/r6/fresh_probe.py $k 40 2>&1 | tail -1 | cut -c1-600
done
bash tools/r6/call18.sh 2>&1 | grep -v "hipLaunchKernel" | sed -n 8,60p
