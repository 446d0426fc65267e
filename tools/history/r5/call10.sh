#!/bin/bash
# round 5, GPU call 10: the general gradient kernel with its walk again (power-law gradient)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c10
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cells.py -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q -k "several_row_windows or sum_then_step or exchange_code_paths" 2>&1 | tail -3
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
timeout 300 python bench.py --zipf 1.1 --no-cpu-baseline --repeats 3 > $O/zipf.json 2> $O/zipf.err; line $O/zipf.json
timeout 400 python bench.py $N8 --repeats 3 --batches 8 --no-owner-leg --key-build-steps 0 > $O/n8_owner.json 2> $O/n8_owner.err; line $O/n8_owner.json
timeout 300 python bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --no-cpu-baseline --repeats 3 --batches 8 > $O/fm64.json 2> $O/fm64.err; line $O/fm64.json
