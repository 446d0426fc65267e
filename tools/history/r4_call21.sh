#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_keybuild.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --no-fm-leg > gpurun_out/bench_lr.json 2> gpurun_out/bench_lr.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_lr.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "value", d["value"], "kernels", d["kernels_ms"])
print("with_kb", json.dumps(d["with_key_build"])[:400])
print("roofline", d["roofline"]["frac"])
PY
tail -3 gpurun_out/bench_lr.err
