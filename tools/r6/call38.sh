#!/bin/bash
# round 6, call 38: the power-law step under the gradient kernel's named paths
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
Q="--no-cpu-baseline --no-fresh-table --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-fm-leg --no-table-sweep --key-build-steps 0 --zipf 1.1 --signal-keys 0 --batches 8"
for t in "lr_gradient=0" "lr_gradient=1" "lr_gradient=2" "lr_gradient=3" "old_weight=1" "old_weight=2"; do
python bench.py $Q --tune $t --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t', round(d['ms_per_step'],4), d['ms_per_step_repeats']['median'] if d.get('ms_per_step_repeats') else None, {k:round(v*1e3,1) for k,v in d['kernels_ms'].items() if v})"
done
