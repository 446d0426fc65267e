#!/bin/bash
# round 5, GPU call 16: 3e7 keys (state 360 MB: past the Infinity Cache, touch density 0.33): the
# dense gradient kernel with the old weights derived (exp_knob 279) against read (default there)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c16
mkdir -p $O
show() {
python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "lr %.4f ms" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["kernels_ms"].items() if v})
    for t in d["table_sweep"]["tables"]:
        if "error" in t:
            print("  ", t); continue
        print("   %.0e keys: %.4f ms" % (t["keys_per_gpu"], t["ms_per_step"]), {k: round(v * 1e3, 1) for k, v in t["kernels_ms"].items() if v},
              "frac %.3f" % t["roofline"]["frac"], "wkb %.3f ms" % t["with_key_build_ms_per_step"])
except Exception as e:
    print(f, "FAILED", e)
PY
}
B="--batches 8 --no-cpu-baseline --no-fm-leg --no-zipf-leg --key-build-steps 0 --repeats 0 --sweep-keys 20000000,30000000,50000000"
for K in 279 0; do
  timeout 600 python bench.py $B --exp-knob $K > $O/sweep_k$K.json 2> $O/sweep_k$K.err; show $O/sweep_k$K.json
done
