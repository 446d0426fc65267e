#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 900 python -m pytest tests/test_gpu_keybuild.py -m gpu -x -q -k "settles or rows_of_16" 2>&1 | tail -8
