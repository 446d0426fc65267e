#!/bin/bash
# round 6, call 9: the first-touch kernels' atomics in flight together; the compact dense variant
# that issues nothing for idle slots, at 3e7 and 1e8 keys
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4
rm -rf /tmp/ft
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -- \
    python "$GRAFT_REPO_ROOT/tools/r6/fresh_probe.py" 10000000 3 30 > /tmp/ft.out 2> /tmp/ft.err)
python tools/r6/timeline.py /tmp/ft 0 60 | grep "k_ar_\|k_kb_scatter\|k_lr_fwd" | cut -c1-100
for nk in 10000000 100000000; do
  python tools/r6/fresh_probe.py $nk 40 30 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fresh', d['keys_per_gpu'], '%.3g ex/s' % d['value'], 'ms/mb %.3f' % d['ms_per_minibatch'], 'first %.2f' % d['ms_first_minibatch'], 'last5 %.3f' % d['ms_last_5_minibatches'], 'defrags', [(x['after_minibatch'], round(x['ms'],2)) for x in d['defrags']]); print('   ', d['ms_by_minibatch'][:20])"
done
timeout 900 python tools/r6/sweep_variants.py 100000000 2>&1 | tail -7 | cut -c1-200
timeout 900 python tools/r6/sweep_variants.py 30000000 2>&1 | tail -7 | cut -c1-200
