#!/bin/bash
# round 5, GPU call 20: the sharded trainer's device arrays from the allocation pool (compile +
# step of fresh minibatches at the N = 8 shard shape); the 10^8-key table once more, the old
# weights derived (default) against read (exp_knob 280), twice each on this box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c20
mkdir -p $O
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --signal-keys 0 --force-sharded --general-path --schedule owner --no-cpu-baseline"
for i in 1 2; do
timeout 400 python bench.py $N8 --repeats 2 --batches 8 --no-owner-leg > $O/n8_owner_$i.json 2> $O/n8_owner_$i.err
python - $O/n8_owner_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("n8 owner %.4f ms" % d["ms_per_step"], "with_key_build", d.get("ms_per_step_with_key_build"), (d.get("with_key_build") or {}).get("from_host_arrays_ms_per_step"))
PY
done
show() {
python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    for t in d["table_sweep"]["tables"]:
        if "error" in t:
            print("  ", t); continue
        print(f.split("/")[-1], "  %.0e keys: %.4f ms" % (t["keys_per_gpu"], t["ms_per_step"]), {k: round(v * 1e3, 1) for k, v in t["kernels_ms"].items() if v},
              "frac %.3f" % t["roofline"]["frac"], "wkb %.3f ms" % t["with_key_build_ms_per_step"])
except Exception as e:
    print(f, "FAILED", e)
PY
}
B="--batches 8 --no-cpu-baseline --no-fm-leg --no-zipf-leg --key-build-steps 0 --repeats 0 --steps 8 --warmup 4 --sweep-keys 100000000"
for K in 0 280 0 280; do
  timeout 600 python bench.py $B --exp-knob $K > $O/sweep_k${K}.json 2> $O/sweep_k$K.err; show $O/sweep_k$K.json
done
