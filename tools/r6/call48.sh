#!/bin/bash
# round 6, call 48: the whole GPU suite on the library with the hand-written sort and the
# worker-side LR build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
