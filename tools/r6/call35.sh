#!/bin/bash
# round 6, call 35 (experiment): 1650 key ranges for a table's first minibatch (the scatter's full
# tiles) against 2442
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 900 python -m pytest tests/test_gpu_keybuild.py -m gpu -x -q -k "settles or rows_of_16 or pushed_first" 2>&1 | tail -2
for k in 10000000 100000000; do
  for i in 1 2; do timeout 600 python tools/r6/fresh_probe.py $k 6 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print($k, 'first', round(d['ms_first_minibatch'],3), [round(x,2) for x in d['ms_by_minibatch']])"; done
done
bash tools/r6/call18.sh 2>&1 | grep "KERN k_eb\|KERN k_kb_sc\|KERN k_kb_hist_g"
