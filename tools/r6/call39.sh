#!/bin/bash
# round 6, call 39: the settle rule's threshold again (30 % since round 6's first measurement),
# now that the first minibatch settles the table by itself
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
for k in 10000000 100000000; do
for pct in 15 30 50 100; do
  timeout 600 python tools/r6/fresh_probe.py $k 40 $pct 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print($k, $pct, '%.3g ex/s' % d['value'], 'defrags', [(x['after_minibatch'], round(x['ms'],1)) for x in d['defrags']], 'last5 %.2f' % d['ms_last_5_minibatches'])"
done; done
