#!/bin/bash
# round 6, call 25: host waits — interrupt-driven (default) against polling signal waits
# (HSA_ENABLE_INTERRUPT=0) on the legs that wait for the GPU several times per minibatch
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%.3g ex/s first %.3f ms steady %.3f ms' % (d['value'], d['ms_first_minibatch'], d['ms_last_5_minibatches']), [round(x,2) for x in d['ms_by_minibatch'][:12]])"; }
echo default;      python tools/r6/fresh_probe.py 10000000 40 2>/dev/null | tail -1 | show
echo polling;      HSA_ENABLE_INTERRUPT=0 python tools/r6/fresh_probe.py 10000000 40 2>/dev/null | tail -1 | show
echo default;      python tools/r6/fresh_probe.py 10000000 40 2>/dev/null | tail -1 | show
echo polling 1e8;  HSA_ENABLE_INTERRUPT=0 python tools/r6/fresh_probe.py 100000000 40 2>/dev/null | tail -1 | show
for m in default polling; do
  [ $m = polling ] && export HSA_ENABLE_INTERRUPT=0
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fresh-table --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-fm-leg --no-zipf-leg --no-table-sweep 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d.get('ms_per_step_with_key_build'))"
done
