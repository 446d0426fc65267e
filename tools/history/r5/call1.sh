#!/bin/bash
# round 5, GPU call 1: the several-workers gradient pass, the owner-compute compile from device
# arrays, the N = 8 shard shape with its key build, the new bench legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -k "several_workers or compile_from_device or owner_compute_dataflow_ranks or several_row_windows or sum_then_step or exchange_code_paths or overlapped" 2>&1 | tail -5
timeout 300 python -m pytest tests/test_gpu_cells.py -x -q 2>&1 | tail -3
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --signal-keys 0 --force-sharded --general-path --schedule owner --no-cpu-baseline --repeats 3 --batches 8"
line() {
python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.4f ms" % d["ms_per_step"], d.get("ms_per_step_repeats") and "median %.4f" % d["ms_per_step_repeats"]["median"],
          {k: round(v * 1e3, 1) for k, v in d["kernels_ms"].items() if v}, "wkb", d.get("with_key_build"))
except Exception as e:
    print(f, "FAILED", e)
PY
}
timeout 400 python bench.py $N8 > $O/n8_owner.json 2> $O/n8_owner.err; line $O/n8_owner.json
for K in 0 297 298; do
XF_OWNER_TIMING_SOURCES=8 timeout 400 python bench.py $N8 --no-owner-leg --key-build-steps 0 --exp-knob $K > $O/n8_src8_k$K.json 2> $O/n8_src8_k$K.err; line $O/n8_src8_k$K.json
done
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c1/bench_n1.json").read().strip().splitlines()[-1])
print("n1 ms/step", d["ms_per_step"], d["kernels_ms"], "frac", d["roofline"]["frac"], "wkb", d.get("ms_per_step_with_key_build"))
z = d.get("zipf", {})
print("zipf", {k: z.get(k) for k in ("ms_per_step", "kernels_ms", "with_key_build_ms_per_step", "error")}, z.get("roofline", {}).get("frac"))
for t in d.get("table_sweep", {}).get("tables", []):
    print("sweep", {k: t.get(k) for k in ("keys_per_gpu", "table_keys", "ms_per_step", "kernels_ms", "with_key_build_ms_per_step", "setup_s", "error")}, t.get("roofline", {}).get("frac"), t.get("cells"))
print("fm", d.get("fm", {}).get("ms_per_step"), (d.get("fm", {}).get("with_key_build") or {}).get("ms_per_step"))
PY
tail -5 $O/bench_n1.err
