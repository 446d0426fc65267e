#!/bin/bash
# round 5, GPU call 8: the several-workers pass with its sums in the stepping lanes' slots
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c8
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q -k "several_workers or owner_compute_dataflow_ranks or several_row_windows" 2>&1 | tail -4
timeout 300 python -m pytest tests/test_gpu_ingest.py -x -q 2>&1 | tail -3
line() {
python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.4f ms" % d["ms_per_step"], d.get("ms_per_step_repeats") and "median %.4f" % d["ms_per_step_repeats"]["median"],
          {k: round(v * 1e3, 1) for k, v in d["kernels_ms"].items() if v})
except Exception as e:
    print(f, "FAILED", e)
PY
}
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --signal-keys 0 --force-sharded --general-path --schedule owner --no-cpu-baseline"
for K in 0 295 0 295; do
XF_OWNER_TIMING_SOURCES=8 timeout 400 python bench.py $N8 --repeats 3 --batches 8 --no-owner-leg --key-build-steps 0 --exp-knob $K > $O/n8_src8_k$K.json 2> $O/n8_src8_k$K.err; line $O/n8_src8_k$K.json
done
timeout 300 python bench.py --zipf 1.1 --no-cpu-baseline --repeats 3 > $O/zipf.json 2> $O/zipf.err; line $O/zipf.json
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/wkb -- \
    python $R/bench.py $N8 --steps 4 --warmup 2 --repeats 0 --batches 2 --no-owner-leg --key-build-steps 16 > $R/$O/wkb.json 2> $R/$O/wkb.err
cd $R
line $O/wkb.json
python - <<'PY'
import csv, glob, json
d = json.loads(open("gpurun_out/r5c8/wkb.json").read().strip().splitlines()[-1])
print("wkb", d.get("with_key_build"))
for f in glob.glob("gpurun_out/r5c8/wkb/**/*kernel_stats.csv", recursive=True)[:1]:
    rows = [r for r in csv.DictReader(open(f)) if "k_kb" in r["Name"] or "k_own" in r["Name"] or "k_lr" in r["Name"] or "k_rows" in r["Name"] or "k_plan" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print("  %-70s calls %6s avg_us %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
