#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c3
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cells.py tests/test_gpu_reforder.py -x -q > $O/t1.log 2>&1; echo "cells+reforder rc=$?" | tee -a $O/summary.txt; tail -15 $O/t1.log | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/t2.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt; tail -15 $O/t2.log | tee -a $O/summary.txt
timeout 300 python tools/cells_knobs.py --knobs 399,300,301,304,300,308,316,332,396,340,300 --steps 32 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
(cd /tmp && timeout 60 rocprofv3 -L > $O/counters.txt 2>&1); wc -l $O/counters.txt | tee -a $O/summary.txt
XF_WORLD8_FM_ROWS=50000 timeout 900 python -m pytest tests/test_gpu_world8_fullsize.py -x -q -k config4 --durations=3 > $O/world8_fm_full.log 2>&1; echo "world8 fm full rows rc=$?" | tee -a $O/summary.txt; tail -8 $O/world8_fm_full.log | tee -a $O/summary.txt
