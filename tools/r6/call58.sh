#!/bin/bash
# round 6, call 58: the first minibatch of the first-epoch leg, the commit of call 49 against HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
(cd _old && python -m xflow_amd.build > /tmp/build_old.log 2>&1 || tail -5 /tmp/build_old.log)
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
for i in 1 2 3; do
for w in _old .; do
  (cd $w && timeout 600 python tools/r6/fresh_probe.py 10000000 12 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', 'first %.3f ms' % d['ms_first_minibatch'], [round(x,2) for x in d['ms_by_minibatch'][:5]])")
done; done
