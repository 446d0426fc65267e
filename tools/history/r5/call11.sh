#!/bin/bash
# round 5, GPU call 11: is the dense gradient kernel's time a staircase in the number of chunks
# (workgroup rounds: 6 per CU x 256 CUs = 1536 chunks per round)?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c11
mkdir -p $O
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
for K in 9400000 10000000 10600000 11000000 11800000 12500000 13000000; do
  timeout 300 python bench.py --keys-per-gpu $K --signal-keys 0 --batches 12 --no-cpu-baseline --no-fm-leg --no-zipf-leg --no-table-sweep --key-build-steps 0 --repeats 2 > $O/k$K.json 2> $O/k$K.err; line $O/k$K.json
done
