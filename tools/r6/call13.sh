#!/bin/bash
# round 6, call 13 (experiment): the dense gradient + Push kernel's first round staggered
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 600 python tools/r6/stagger_probe.py 2>&1 | tail -14
