#!/bin/bash
# round 6, call 12 (investigation): k_ar_insert with parts switched off — which part is the 0.9 ms
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
for dbg in 0 1 2 6 14 8 4; do
  rm -rf /tmp/ft
  (cd /tmp && XF_AR_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -- \
      python "$GRAFT_REPO_ROOT/tools/r6/fresh_probe.py" 10000000 1 3000 > /tmp/ft.out 2> /tmp/ft.err)
  echo "== XF_AR_DEBUG=$dbg"
  python tools/r6/timeline.py /tmp/ft 0 70 | grep "k_ar_insert\|k_ar_place" | tail -3 | cut -c1-100
done
