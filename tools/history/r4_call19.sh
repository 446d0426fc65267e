#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fm_keybuild.py tests/test_gpu_parity.py tests/test_gpu_parity_tight.py -x -q 2>&1 | tail -4
bash tools/r4_fmprof.sh 2>&1 | head -24
