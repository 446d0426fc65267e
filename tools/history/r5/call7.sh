#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
bash tools/profile_round5.sh gpurun_out/r05 2>&1 | tail -120
