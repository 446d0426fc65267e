#!/bin/bash
# round 6, call 28: k_kb_resolve on a power-law stream — a heavy super-chunk's items take their
# cells' slots with one atomic per cell and round (LDS counts) instead of one per wavefront
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
Q="--no-cpu-baseline --no-fresh-table --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-fm-leg --no-table-sweep"
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_z -- python $GRAFT_REPO_ROOT/bench.py $Q --zipf 1.1 --signal-keys 0 --batches 8 --repeats 0 > /tmp/_z.out 2>&1)
grep "k_kb_resolve\|k_kb_scatter\|k_kb_hist" $(find /tmp/_z -name "*kernel_stats.csv" | head -1) | cut -c1-120
tail -1 /tmp/_z.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('zipf line', d['ms_per_step'], d.get('ms_per_step_with_key_build'))"
python bench.py $Q --no-zipf-leg --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('uniform', d['ms_per_step'], d.get('ms_per_step_with_key_build'))"
