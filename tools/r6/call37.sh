#!/bin/bash
# round 6, call 37: the 10^8-key first-epoch leg four times (twice this round a run had defrags
# of 100 ms: calls 19 and 36), the arena now sized for the defrags
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
for i in 1 2 3 4; do
  timeout 600 python tools/r6/fresh_probe.py 100000000 40 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%.3g ex/s first %.2f' % (d['value'], d['ms_first_minibatch']), 'max %.1f' % max(d['ms_by_minibatch']), [round(x['ms'],1) for x in d['defrags']])"
done
timeout 600 python tools/r6/fresh_probe.py 10000000 40 2>&1 | tail -1 | cut -c1-200
