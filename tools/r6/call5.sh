#!/bin/bash
# round 6, call 5: the miss list without same-address atomics, the defrag with ordered lists +
# cluster fix + ping-pong state buffers: parity, then the first-epoch leg under three policies
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_parity.py \
  tests/test_gpu_fullsize.py tests/test_gpu_fm_keybuild.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r6/call5_tests.log
cat gpurun_out/r6/call5_tests.log
python bench.py --no-fm-leg --no-zipf-leg --no-table-sweep --no-fresh-table --no-n8-shape --no-end-to-end --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('main', round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['kernels_ms'].items() if v}, 'wkb', d['with_key_build']['ms_per_step'], d['with_key_build']['ms_per_step_repeats'])"
for pct in 30 10 3; do
  python tools/r6/fresh_probe.py 10000000 40 $pct 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('1e7 pct', $pct, '%.3g ex/s' % d['value'], 'ms/mb %.3f' % d['ms_per_minibatch'], 'first %.2f' % d['ms_first_minibatch'], 'last5 %.3f' % d['ms_last_5_minibatches'], 'defrags', [(x['after_minibatch'], round(x['ms'],2)) for x in d['defrags']]); print('   ', d['ms_by_minibatch'][:16])"
done
for pct in 30 10; do
  python tools/r6/fresh_probe.py 100000000 40 $pct 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('1e8 pct', $pct, '%.3g ex/s' % d['value'], 'ms/mb %.3f' % d['ms_per_minibatch'], 'first %.2f' % d['ms_first_minibatch'], 'last5 %.3f' % d['ms_last_5_minibatches'], 'defrags', [(x['after_minibatch'], round(x['ms'],2)) for x in d['defrags']]); print('   ', d['ms_by_minibatch'][:16])"
done
rm -rf /tmp/ft
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d /tmp/ft -- \
    python "$GRAFT_REPO_ROOT/tools/r6/fresh_probe.py" 100000000 10 30 > /tmp/ft.out 2> /tmp/ft.err)
for kind in kernel_stats hip_api_stats; do
  f=$(find /tmp/ft -name "*${kind}.csv" | head -1)
  echo "== 1e8 $kind"
  [ -n "$f" ] && cp "$f" gpurun_out/r6/fresh_1e8_${kind}.csv && python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    name = r.get("Name", "")
    m = re.search(r"k_\w+(<[^>]*>)?", name)
    print("%-52s calls %6s total %10.1f us avg %9.1f us %6s%%" % ((m.group(0) if m else name)[:52], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
