#!/bin/bash
# Experiment (GPU box): library variants A/B on the default LR bench line (no CPU baseline, no
# FM leg).   tools/lr_ab.sh default v1 v2 ...     (EXTRA="--zipf 1.1" etc.)
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ $v = default ]; then unset XF_LIB; else export XF_LIB=$R/xflow_amd/lib/var_$v/libxflow_amd.so; fi
  python $R/bench.py --no-cpu-baseline --no-fm-leg --repeats 4 --batches 4 $EXTRA 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    print('$v', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_repeats']['median'],4), {k: round(v*1e3,1) for k,v in d['kernels_ms'].items() if v}, 'wkb', d.get('ms_per_step_with_key_build'))
"
done
