#!/bin/bash
# round 5, GPU call 21: the whole GPU suite + smoke on the final library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 2>&1 | tail -28
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
