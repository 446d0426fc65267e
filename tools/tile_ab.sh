#!/bin/bash
# Experiment (GPU box): library variants A/B (xflow_amd/lib/var_<v>, tools/build_variant.sh) —
# FM leg (k = 16 + SGD), and with FM64=1 FM k = 64 + FTRL power-law.   tools/tile_ab.sh default v1 v2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    if 'kernels_ms' in d: print('$1', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels_ms'].items()})
"; }
for v in "$@"; do
  if [ $v = default ]; then unset XF_LIB; else export XF_LIB=$R/xflow_amd/lib/var_$v/libxflow_amd.so; fi
  python $R/tools/fm_leg.py --batches 4 2>/dev/null | show "fm16 $v"
  [ -n "$FM64" ] && python $R/bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --steps 12 --warmup 4 --no-cpu-baseline --batches 4 --repeats 2 2>/dev/null | show "fm64-zipf $v"
done
