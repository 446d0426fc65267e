#!/bin/bash
# round 5, GPU call 6: the whole GPU suite and the smoke test on the final code
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -22
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
