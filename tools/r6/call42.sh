#!/bin/bash
# round 6, call 42 (experiment, withdrawn): xf_lr_update_dev sized a workspace that has to grow on a
# second host thread while the build ran; first minibatch 0.84-0.87 ms (10^7 keys) / 1.00-1.04 ms (10^8)
# against 0.80-0.95 / 0.95-1.09 without it: no gain, the code is not in the tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
for k in 10000000 100000000; do
  timeout 600 python tools/r6/fresh_probe.py $k 40 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print($k, '%.3g ex/s first %.3f ms' % (d['value'], d['ms_first_minibatch']), [round(x,2) for x in d['ms_by_minibatch'][:4]])"
done; done
