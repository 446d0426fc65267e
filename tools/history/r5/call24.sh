#!/bin/bash
# round 5, GPU call 24: two workers on one GPU with ingest=gpu (both dataflows)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q 2>&1 | tail -30
