#!/bin/bash
# round 6, call 52: the round's evidence again on the last library (bench lines, kernel stats; the
# PMC passes of profiles/r06 stay those of commit 5b4018f: no kernel of theirs changed since)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
bash tools/profile_round6.sh gpurun_out/r06e nopmc 2>&1 | cut -c1-900
