#!/bin/bash
# round 6, call 55: the first-epoch leg behind the main run, the session's first commit (28f8100)
# against HEAD: are the slow defrags at 1e8 keys new?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
for t in d['fresh_table']:
    print('$1', t['keys_per_gpu'], '%.3g ex/s' % t['value'], 'defrags', [round(x['ms'],1) for x in t['defrags']])"; }
F="--no-cpu-baseline --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-fm-leg --no-zipf-leg --no-table-sweep --key-build-steps 0 --repeats 0"
(cd _old && python -m xflow_amd.build > /tmp/build_old.log 2>&1 || tail -5 /tmp/build_old.log)
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
for i in 1 2; do
(cd _old && python bench.py $F 2>/dev/null | tail -1 | show old)
python bench.py $F 2>/dev/null | tail -1 | show head
done
