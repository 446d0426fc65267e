#!/bin/bash
# round 6, call 16: an empty table's first minibatch settled at once (k_eb_*): tests, then the
# first-epoch leg and its kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
mkdir -p gpurun_out/r06c
for k in 10000000 100000000; do
  timeout 600 python tools/r6/fresh_probe.py $k 40 2>&1 | tail -4 | cut -c1-1500
done
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_f7 -- python $GRAFT_REPO_ROOT/tools/r6/fresh_probe.py 10000000 8 > /tmp/_f7.out 2>&1)
cp $(find /tmp/_f7 -name "*kernel_stats.csv" | head -1) gpurun_out/r06c/fresh_table_1e7_first8_kernel_stats.csv
head -25 gpurun_out/r06c/fresh_table_1e7_first8_kernel_stats.csv | cut -c1-200
