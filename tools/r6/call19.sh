#!/bin/bash
# round 6, call 19: the rank kernels launched before the table's count is known, a range's keys
# kept in registers between the two passes, the first cells' allocation set aside: tests of the
# build, the first-epoch leg, the first minibatch's timeline again
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_ingest.py -m gpu -x -q 2>&1 | tail -5
for k in 10000000 100000000; do
  timeout 600 python tools/r6/fresh_probe.py $k 40 2>&1 | tail -1 | cut -c1-700
done
bash tools/r6/call18.sh 2>&1 | grep -v "hipLaunchKernel" | head -75
