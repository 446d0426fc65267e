#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_parity.py tests/test_gpu_keybuild.py -x -q 2>&1 | tail -2
timeout 600 python bench.py --zipf 1.1 --no-cpu-baseline --no-fm-leg > gpurun_out/bench_zipf.json 2> gpurun_out/bench_zipf.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_zipf.json").read().strip().splitlines()[-1])
print("zipf ms/step", d["ms_per_step"], "value", d["value"], "kernels", d["kernels_ms"], "with_kb", d["ms_per_step_with_key_build"])
print("step frac of peak", d["step_gbs_survey_8d"]/8000.0, "roofline", d["roofline"]["frac"])
PY
