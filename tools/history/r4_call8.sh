#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c8
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cells.py tests/test_gpu_sharded.py -x -q > $O/t1.log 2>&1; echo "cells+sharded rc=$?" | tee -a $O/summary.txt; tail -4 $O/t1.log | tee -a $O/summary.txt
for v in default cb12; do
  if [ $v = default ]; then unset XF_LIB; else export XF_LIB=$R/xflow_amd/lib/var_$v/libxflow_amd.so; fi
  echo "=== $v ===" | tee -a $O/summary.txt
  timeout 300 python tools/cells_knobs.py --knobs 299,0 --steps 32 --zipf 1.1 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
  timeout 300 python tools/cells_knobs.py --knobs 299,0,428 --steps 32 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
  echo "signal keys 32:" | tee -a $O/summary.txt
  timeout 300 python tools/cells_knobs.py --knobs 299,0,428 --steps 32 --signal-keys 32 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
unset XF_LIB
python tools/kb_prof.sh 2>/dev/null | head -0
timeout 300 python bench.py --no-cpu-baseline --no-fm-leg --repeats 3 --batches 8 --signal-keys 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    print('bench signal0', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_repeats']['median'],4), {k: round(v*1e3,1) for k,v in d['kernels_ms'].items() if v}, 'wkb', d.get('ms_per_step_with_key_build'))
" | tee -a $O/summary.txt
timeout 300 python bench.py --no-cpu-baseline --no-fm-leg --repeats 3 --batches 8 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    print('bench signal32', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_repeats']['median'],4), {k: round(v*1e3,1) for k,v in d['kernels_ms'].items() if v}, 'wkb', d.get('ms_per_step_with_key_build'), d['with_key_build'].get('ms_per_step_repeats'))
" | tee -a $O/summary.txt
