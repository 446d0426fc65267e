#!/bin/bash
# round 6, call 1: the first-touch build (k_ar_*), the hand-written cell sort, the named path
# switches — parity suites that cover them, then the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_parity.py \
  tests/test_gpu_reforder.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r6/call1_tests.log
cat gpurun_out/r6/call1_tests.log
timeout 600 python bench.py --no-table-sweep --no-fm-leg > gpurun_out/r6/call1_bench.json 2> gpurun_out/r6/call1_bench.err
tail -c 3000 gpurun_out/r6/call1_bench.json; tail -5 gpurun_out/r6/call1_bench.err
