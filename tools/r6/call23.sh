#!/bin/bash
# round 6, call 23: k_eb_rank without the cluster sort (a rank = a walk over the cluster)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_keybuild.py -m gpu -x -q 2>&1 | tail -3
for k in 10000000 100000000; do
  timeout 600 python tools/r6/fresh_probe.py $k 40 2>&1 | tail -1 | cut -c1-420
done
bash tools/r6/call18.sh 2>&1 | grep "KERN k_eb\|KERN k_kb_sc"
