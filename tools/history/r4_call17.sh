#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
timeout 600 python tools/cells_knobs.py --signal-keys 32 --knobs 0,299 2>&1 | tail -3
timeout 600 python tools/cells_knobs.py --zipf 1.1 --signal-keys 32 --knobs 0 --batches 16 2>&1 | tail -2
timeout 600 python tools/kb_knobs.py --knobs 0 --iters 8 --step 2>&1 | tail -6
