#!/bin/bash
# round 6, call 53: the official K-step block against its repeats, eight runs (one of the round's
# profile runs had 1.88 ms per step in the official block, 0.109 in the ten blocks after it)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
QUIET="--no-cpu-baseline --no-fresh-table --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-fm-leg --no-zipf-leg --no-table-sweep --key-build-steps 0"
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 $QUIET 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['ms_per_step_repeats']
print('run $i: official %.4f ms  repeats median %.4f min %.4f max %.4f' % (d['ms_per_step'], r['median'], r['min'], r['max']))"
done
