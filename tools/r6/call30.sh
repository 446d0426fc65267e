#!/bin/bash
# round 6, call 30: the FM key build on the power-law stream (fm_with_key_build 5.5 ms in the
# Zipf bench line against 1.3 ms on the uniform one): which kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
Q="--no-cpu-baseline --no-fresh-table --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-zipf-leg --no-table-sweep --key-build-steps 0 --repeats 0"
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_f -- python $GRAFT_REPO_ROOT/bench.py $Q --zipf 1.1 > /tmp/_f.out 2>&1)
F=$(find /tmp/_f -name "*kernel_stats.csv" | head -1); head -22 $F | cut -c1-150; grep "k_eb_" $F | cut -c1-150
tail -1 /tmp/_f.out | cut -c1-300
