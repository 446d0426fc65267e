#!/bin/bash
# round 5, GPU call 14: the gradient + Push kernels derive the old weight from (n, z)
# (TableDev::w_of_nz) instead of reading it; exp_knob 280 = they read it (A/B)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c14
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cells.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -k "several or sum_then_step or exchange_code_paths or rank_ordered" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
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
B="--batches 12 --no-cpu-baseline --no-fm-leg --no-zipf-leg --no-table-sweep --key-build-steps 0 --repeats 2"
for K in 0 280 0 280; do
  timeout 300 python bench.py $B --exp-knob $K > $O/lr_k$K.json 2> $O/lr_k$K.err; line $O/lr_k$K.json
done
timeout 300 python bench.py --zipf 1.1 --no-cpu-baseline --repeats 2 > $O/zipf.json 2> $O/zipf.err; line $O/zipf.json
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --signal-keys 0 --force-sharded --general-path --schedule owner --no-cpu-baseline"
timeout 400 python bench.py $N8 --repeats 2 --batches 8 --no-owner-leg --key-build-steps 0 > $O/n8_owner.json 2> $O/n8_owner.err; line $O/n8_owner.json
XF_OWNER_TIMING_SOURCES=8 timeout 400 python bench.py $N8 --repeats 2 --batches 8 --no-owner-leg --key-build-steps 0 > $O/n8_src8.json 2> $O/n8_src8.err; line $O/n8_src8.json
