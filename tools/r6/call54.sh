#!/bin/bash
# round 6, call 54: the first-epoch leg at 1e8 keys alone and behind the other legs (its defrags:
# 3-5 ms alone; 14-72 ms in one full bench run)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
ts = d['fresh_table'] if 'fresh_table' in d else [d]
for t in ts:
    print('$1', t['keys_per_gpu'], '%.3g ex/s' % t['value'], 'defrags', [round(x['ms'],1) for x in t['defrags']])"; }
python tools/r6/fresh_probe.py 100000000 40 2>/dev/null | tail -1 | show probe
python bench.py --no-cpu-baseline --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-fm-leg --no-zipf-leg --no-table-sweep --key-build-steps 0 --repeats 0 2>/dev/null | tail -1 | show only_fresh
python bench.py --no-cpu-baseline --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-zipf-leg --no-table-sweep --key-build-steps 0 --repeats 0 2>/dev/null | tail -1 | show fm_then_fresh
python bench.py --no-cpu-baseline --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-fm-leg --no-zipf-leg --key-build-steps 0 --repeats 0 2>/dev/null | tail -1 | show sweep_then_fresh
