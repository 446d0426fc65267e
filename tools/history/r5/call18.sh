#!/bin/bash
# round 5, GPU call 18: k_lr_grad_ranked (the workers' phases merged) against k_lr_grad_multi
# (exp_knob 293): parity tests, then the N = 8 owner shape with 4 / 8 / 16 pretended sources
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c18
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q -k "several or rank_ordered or exchange_code_paths or owner" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_world8_fullsize.py -x -q 2>&1 | tail -3
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
for S in 8 2 4 16; do
for K in 0 293; do
XF_OWNER_TIMING_SOURCES=$S timeout 400 python bench.py $N8 --repeats 2 --batches 8 --no-owner-leg --key-build-steps 0 --exp-knob $K > $O/n8_src${S}_k$K.json 2> $O/n8_src${S}_k$K.err; line $O/n8_src${S}_k$K.json
done
done
