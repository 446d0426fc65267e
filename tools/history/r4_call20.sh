#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fm_keybuild.py tests/test_gpu_sharded.py tests/test_gpu_cli.py tests/test_gpu_parity_tight.py -x -q 2>&1 | tail -15
