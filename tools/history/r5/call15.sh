#!/bin/bash
# round 5, GPU call 15: tables beyond the Infinity Cache (3e7 / 1e8 keys): the dense gradient
# kernel deriving the old weights from (n, z) where the minibatch touches the chunks thinly
# (default) against reading w (exp_knob 280)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c15
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cells.py -x -q -k "old_weight or every_variant" 2>&1 | tail -3
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
B="--batches 8 --no-cpu-baseline --no-fm-leg --no-zipf-leg --key-build-steps 0 --repeats 0"
for K in 0 280; do
  timeout 600 python bench.py $B --exp-knob $K > $O/sweep_k$K.json 2> $O/sweep_k$K.err; show $O/sweep_k$K.json
done
