#!/bin/bash
# round 5, GPU call 27: windows at the N = 8 owner shape staged on one rank (23 .. 48; the rule picks
# 32) through an XF_OWNER_WINDOWS override that existed for this call only (kept for the record)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c27
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
for NW in 23 26 32 40 48; do
XF_OWNER_WINDOWS=$NW XF_OWNER_TIMING_SOURCES=8 timeout 400 python bench.py $N8 --repeats 2 --batches 8 --no-owner-leg --key-build-steps 0 > $O/n8_src8_w$NW.json 2> $O/n8_src8_w$NW.err; line $O/n8_src8_w$NW.json
done
