#!/bin/bash
# round 6, call 31: the FM key build's count / regroup over work items (a heavy super-chunk's
# records are many workgroups' work): tests, the power-law stream again
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_fm_keybuild.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
bash tools/r6/call30.sh 2>&1 | grep "k_fm_regroup\|k_fm_count\|k_kb_resolve_fm\|k_eb_rank" | cut -c1-140
Q="--no-cpu-baseline --no-fresh-table --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-zipf-leg --no-table-sweep"
for z in "" "--zipf 1.1"; do
python bench.py $Q $z --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['summary']; print('$z', s['ms_per_step'], s['with_key_build_ms'], s.get('fm_ms_per_step'), s.get('fm_with_key_build_ms'))"
done
