#!/bin/bash
# round 6, call 10: the arrival build with ranges of 4096 records (their index slices fit an XCD's L2)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py -x -q -m gpu 2>&1 | tail -3
rm -rf /tmp/ft
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -- \
    python "$GRAFT_REPO_ROOT/tools/r6/fresh_probe.py" 10000000 3 30 > /tmp/ft.out 2> /tmp/ft.err)
python tools/r6/timeline.py /tmp/ft 0 60 | grep "k_ar_\|k_kb_scatter\|k_kb_hist\|k_lr_fwd" | cut -c1-100
for nk in 10000000 100000000; do
  python tools/r6/fresh_probe.py $nk 40 30 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fresh', d['keys_per_gpu'], '%.3g ex/s' % d['value'], 'ms/mb %.3f' % d['ms_per_minibatch'], 'first %.2f' % d['ms_first_minibatch'], 'last5 %.3f' % d['ms_last_5_minibatches'], 'defrags', [(x['after_minibatch'], round(x['ms'],2)) for x in d['defrags']]); print('   ', d['ms_by_minibatch'][:12])"
done
