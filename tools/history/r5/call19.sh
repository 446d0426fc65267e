#!/bin/bash
# round 5, GPU call 19: the round's evidence on the final kernels (bench lines, kernel stats, PMC)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
time bash tools/profile_round5.sh gpurun_out/r05b 2>&1 | tail -60
