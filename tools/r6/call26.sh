#!/bin/bash
# round 6, call 26: the host-side parts of a minibatch of the first-epoch leg
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
python tools/r6/leg_parts.py 2>/dev/null | tail -30
