#!/bin/bash
# round 5, GPU call 17: the several-workers pass against the number of workers (2, 4, 8, 16
# pretended sources over the same 32 windows): what a phase costs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c17
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
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --signal-keys 0 --force-sharded --general-path --schedule owner --no-cpu-baseline"
for S in 2 4 8 16 32; do
XF_OWNER_TIMING_SOURCES=$S timeout 400 python bench.py $N8 --repeats 2 --batches 8 --no-owner-leg --key-build-steps 0 > $O/n8_src$S.json 2> $O/n8_src$S.err; line $O/n8_src$S.json
done
