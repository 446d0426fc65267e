#!/bin/bash
# round 6, call 22 (an experiment build, not kept): the phases of k_eb_rank by the wall clock
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 600 python tools/r6/fresh_probe.py 10000000 2 2>&1 | grep EBT
timeout 600 python tools/r6/fresh_probe.py 100000000 2 2>&1 | grep EBT
