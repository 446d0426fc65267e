#!/bin/bash
# Experiment (GPU box): gradient tile size A/B — FM leg, the LR exchange path's stages on one
# GPU, FM k=64 + FTRL power-law.  Libraries: default and xflow_amd/lib/var_<v> (build_variant.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    if 'kernels_ms' in d: print('$1', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels_ms'].items()})
"; }
for v in default g128 g2048; do
  if [ $v = default ]; then unset XF_LIB; else export XF_LIB=$R/xflow_amd/lib/var_$v/libxflow_amd.so; fi
  python $R/tools/fm_leg.py --batches 4 2>/dev/null | show "fm16 $v"
  [ $v = g128 ] && continue
  python $R/bench.py --force-sharded --general-path --schedule sequential --steps 20 --warmup 4 --no-cpu-baseline --batches 4 --key-build-steps 0 --repeats 2 2>/dev/null | show "lr-exchange $v"
  python $R/bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --steps 12 --warmup 4 --no-cpu-baseline --batches 4 --repeats 2 2>/dev/null | show "fm64-zipf $v"
done
