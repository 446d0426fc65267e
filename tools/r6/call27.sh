#!/bin/bash
# round 6, call 27: the round's profiles on the final kernels (tools/profile_round6.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
bash tools/profile_round6.sh gpurun_out/r06 2>&1 | cut -c1-600
du -sh gpurun_out/r06; ls gpurun_out/r06 | wc -l
