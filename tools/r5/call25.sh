#!/bin/bash
# round 5, GPU call 25: k_lr_finalize_cells with 16 lanes per row (default where G >= 16) against
# 4 (exp_knob 289)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c25
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "not fm" 2>&1 | tail -3
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
B="--batches 12 --no-cpu-baseline --no-fm-leg --no-zipf-leg --no-table-sweep --key-build-steps 0 --repeats 3"
for K in 0 289 0 289; do
  timeout 300 python bench.py $B --exp-knob $K > $O/lr_k$K.json 2> $O/lr_k$K.err; line $O/lr_k$K.json
done
