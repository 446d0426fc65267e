#!/bin/bash
# round 6, call 20: the 10^8-key first-epoch leg twice (three of its defrags took 100 ms in call 19)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
for i in 1 2; do
  timeout 600 python tools/r6/fresh_probe.py 100000000 40 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_first_minibatch'], [round(x,1) for x in d['ms_by_minibatch']])
print(d['defrags'])"
done
