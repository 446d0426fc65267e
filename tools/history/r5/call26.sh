#!/bin/bash
# round 5, GPU call 26: the whole GPU suite and the end-to-end run on the round's last library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c26
mkdir -p $O
time timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python tools/e2e_text.py 1200000 $O/e2e.json 2>&1 | tail -3
