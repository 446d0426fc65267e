#!/bin/bash
# round 6, call 17: the first build settles a table that holds only what the host API pushed
# (key 0); one allocation per tier, the directories in one kernel: the whole GPU suite, the
# allocator's latencies, the first-epoch leg
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
python tools/r6/malloc_probe.py 2>&1 | tail -8
for k in 10000000 100000000; do
  timeout 600 python tools/r6/fresh_probe.py $k 40 2>&1 | tail -1 | cut -c1-1300
done
mkdir -p gpurun_out/r06c
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_f7 -- python $GRAFT_REPO_ROOT/tools/r6/fresh_probe.py 10000000 8 > /tmp/_f7.out 2>&1)
cp $(find /tmp/_f7 -name "*kernel_stats.csv" | head -1) gpurun_out/r06c/fresh_table_1e7_first8_kernel_stats.csv
grep "k_eb\|k_build_dirs\|k_early\|k_kb_scatter<false, 4096\|hist_groups<false" gpurun_out/r06c/fresh_table_1e7_first8_kernel_stats.csv | cut -c1-160
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
