#!/bin/bash
# round 6, call 3: the split cells files + ADVICE fixes under the suites that cover them; where
# the first-epoch leg's milliseconds go (HIP API + kernel trace); the dense kernel's variants on
# a 1e8-key table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_ingest.py \
  tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6/call3_tests.log
cat gpurun_out/r6/call3_tests.log
for pct in 30 2; do
  python tools/r6/fresh_probe.py 10000000 14 $pct 2>&1 | tail -1 | cut -c1-1200
done
cd /tmp && rocprofv3 --kernel-trace --hip-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r6/fresh_trace" -o fresh -- \
  python "$GRAFT_REPO_ROOT/tools/r6/fresh_probe.py" 10000000 14 30 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/r6/fresh_trace -name "*stats*" | head; 
for f in $(find gpurun_out/r6/fresh_trace -name "*hip_api_stats.csv" -o -name "*kernel_stats.csv"); do echo "== $f"; head -25 "$f" | cut -c1-160; done
find gpurun_out/r6/fresh_trace -name "*trace.csv" -size +20M -delete
timeout 900 python tools/r6/sweep_variants.py 100000000 2>&1 | tail -9 | cut -c1-400
python tools/r6/fresh_probe.py 100000000 10 30 2>&1 | tail -1 | cut -c1-1500
