#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fm_keybuild.py -x -q 2>&1 | tail -25
bash tools/r4_fmprof.sh 2>&1 | head -34
