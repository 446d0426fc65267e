#!/bin/bash
# round 6, call 2: the hand-written defrag (k_df_*) under the whole GPU suite, then the bench
# line with its new legs (sustained, fresh_table, n8_shape, end_to_end)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r6/call2_tests.log
cat gpurun_out/r6/call2_tests.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6/call2_bench.json 2> gpurun_out/r6/call2_bench.err ) 2>&1 | tail -4
tail -c 2500 gpurun_out/r6/call2_bench.json; tail -5 gpurun_out/r6/call2_bench.err
