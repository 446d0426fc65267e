#!/bin/bash
# round 5, GPU call 12: k_lr_grad_multi with two mark arrays (two barriers per phase, exp_knob 294)
# against the three-barrier phases; the worker's tokeniser buffers at start-up (e2e)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c12
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -k "several_workers_pass" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_ingest.py -x -q 2>&1 | tail -3
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
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --signal-keys 0 --force-sharded --general-path --schedule owner --no-cpu-baseline"
for K in 0 294 0 294; do
XF_OWNER_TIMING_SOURCES=8 timeout 400 python bench.py $N8 --repeats 3 --batches 8 --no-owner-leg --key-build-steps 0 --exp-knob $K > $O/n8_src8_k$K.json 2> $O/n8_src8_k$K.err; line $O/n8_src8_k$K.json
done
timeout 900 python tools/e2e_text.py 1200000 $O/e2e.json 2>&1 | tail -12
