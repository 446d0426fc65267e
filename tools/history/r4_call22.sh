#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --no-fm-leg --repeats 2 > gpurun_out/bench_lr.json 2> gpurun_out/bench_lr.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_lr.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "kernels", d["kernels_ms"])
w=d["with_key_build"]
print("with_kb", w["ms_per_step"], "two calls", w["two_calls_ms_per_step"], "piped", json.dumps(w["next_build_under_this_step"]))
PY
tail -3 gpurun_out/bench_lr.err
timeout 300 python -m pytest tests/test_gpu_keybuild.py -x -q 2>&1 | tail -2
