#!/bin/bash
# round 5, GPU call 5: the GPU tokeniser — tests, then end to end through xflow_lr
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c5
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ingest.py -x -q --durations=5 2>&1 | tail -40
E2E_TRACE=1 timeout 900 python tools/e2e_text.py 1200000 $O/e2e.json 2>&1 | tail -40
