#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c2
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_cells.py -x -q > $O/cells_tests.log 2>&1; echo "cells tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/cells_tests.log
timeout 300 python tools/cells_knobs.py --knobs 399,300,301,302,304,305,306,399,301,300 --steps 32 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout 300 python tools/cells_knobs.py --knobs 399,300,301,305 --steps 32 --zipf 1.1 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
if [ "$1" = full ]; then
  timeout 1500 python -m pytest tests/test_gpu_world8_fullsize.py -x -q --durations=5 > $O/world8_full.log 2>&1; echo "world8 full rc=$?" | tee -a $O/summary.txt; tail -40 $O/world8_full.log | tee -a $O/summary.txt
fi
