#!/bin/bash
# round 6, call 36: a heavy range in parts (k_eb_rank<false> per part, <true> merges): tests, the
# power-law first build's kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
bash tools/r6/call30.sh 2>&1 | grep "k_eb_\|k_kb_scatter<false, 4096" | cut -c1-150
for k in 10000000 100000000; do
  timeout 600 python tools/r6/fresh_probe.py $k 40 2>&1 | tail -1 | cut -c1-260
done
