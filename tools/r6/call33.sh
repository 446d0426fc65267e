#!/bin/bash
# round 6, call 33: what the driver runs at the round's end — build(), smoke(), the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
( time python bench.py ) 2>&1 | tail -5 | cut -c1-600
