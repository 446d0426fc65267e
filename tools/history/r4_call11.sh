#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c11
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q > $O/t1.log 2>&1; echo "sharded rc=$?" | tee -a $O/summary.txt; tail -5 $O/t1.log | tee -a $O/summary.txt
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --signal-keys 0 --no-cpu-baseline --repeats 3 --batches 8 --no-owner-leg --heldout-rows 20000"
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    print('$1', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_repeats']['median'],4), {k: round(v*1e3,1) for k,v in d['kernels_ms'].items() if v}, d.get('owner_compute_without_overlap_ms_per_step'))
"; }
python bench.py $N8 --force-sharded --general-path --schedule owner 2>$O/e1 | line owner | tee -a $O/summary.txt
python bench.py $N8 --force-sharded --general-path --schedule owner_stale1 2>$O/e2 | line owner_stale1 | tee -a $O/summary.txt
XF_OWNER_TIMING_SOURCES=8 python bench.py $N8 --force-sharded --general-path --schedule owner 2>$O/e3 | line owner_8src | tee -a $O/summary.txt
XF_OWNER_TIMING_SOURCES=8 python bench.py $N8 --force-sharded --general-path --schedule owner_stale1 2>$O/e4 | line owner_stale1_8src | tee -a $O/summary.txt
python bench.py $N8 2>$O/e5 | line fused | tee -a $O/summary.txt
tail -3 $O/e2
