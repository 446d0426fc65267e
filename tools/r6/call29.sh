#!/bin/bash
# round 6, call 29: the slices per chunk computed by many workgroups (k_kb_psum) instead of the
# scan's plan workgroup: tests, the 10^8-key build, the N = 8 shape's compile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fm_keybuild.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -3
Q="--no-cpu-baseline --no-fresh-table --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-fm-leg --no-zipf-leg"
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_s -- python $GRAFT_REPO_ROOT/bench.py $Q --steps 4 --warmup 2 --repeats 0 --batches 2 --sweep-keys 100000000 > /tmp/_s.out 2>&1)
grep "k_kb_scan\|k_kb_psum\|k_kb_resolve" $(find /tmp/_s -name "*kernel_stats.csv" | head -1) | cut -c1-150
python bench.py $Q --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('line', d['ms_per_step'], d.get('ms_per_step_with_key_build'), d.get('summary'))"
