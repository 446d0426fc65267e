#!/bin/bash
# round 5, GPU call 13: the dense gradient kernel at 5 / 6 / 7 workgroups per CU instead of 8
# (4883 chunks: 2.38 rounds of 2048 — do fewer slots with evener rounds finish sooner?)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c13
mkdir -p $O
line() {
python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.4f ms" % d["ms_per_step"], d.get("ms_per_step_repeats") and "median %.4f" % d["ms_per_step_repeats"]["median"],
          {k: round(v * 1e3, 1) for k, v in d["kernels_ms"].items() if v}, "frac", round(d["roofline"]["frac"], 3))
except Exception as e:
    print(f, "FAILED", e)
PY
}
for K in 0 285 286 287 284 0; do
  timeout 300 python bench.py --batches 12 --no-cpu-baseline --no-fm-leg --no-zipf-leg --no-table-sweep --key-build-steps 0 --repeats 2 --exp-knob $K > $O/k$K.json 2> $O/k$K.err; line $O/k$K.json
done
