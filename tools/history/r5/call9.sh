#!/bin/bash
# round 5, GPU call 9: the loads back under the lanes' masks, the several-workers pass with its
# sums in slots as the default
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c9
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cells.py tests/test_gpu_keybuild.py -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q -k "several_workers or owner_compute_dataflow_ranks or several_row_windows or overlapped" 2>&1 | tail -3
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
timeout 300 python bench.py --no-cpu-baseline --no-fm-leg --no-table-sweep --repeats 5 > $O/bench_lr.json 2> $O/bench_lr.err; line $O/bench_lr.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c9/bench_lr.json").read().strip().splitlines()[-1])
z = d.get("zipf", {}); print("zipf leg", z.get("ms_per_step"), z.get("kernels_ms"))
PY
timeout 300 python bench.py --zipf 1.1 --no-cpu-baseline --repeats 3 > $O/zipf.json 2> $O/zipf.err; line $O/zipf.json
XF_OWNER_TIMING_SOURCES=8 timeout 400 python bench.py $N8 --repeats 3 --batches 8 --no-owner-leg --key-build-steps 0 > $O/n8_src8.json 2> $O/n8_src8.err; line $O/n8_src8.json
timeout 400 python bench.py $N8 --repeats 3 --batches 8 --no-owner-leg --key-build-steps 0 > $O/n8_owner.json 2> $O/n8_owner.err; line $O/n8_owner.json
