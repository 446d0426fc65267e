#!/bin/bash
# round 5, GPU call 4: the fixes of call 3 (group histogram, row-id histogram, the 3.6e7-key
# test), the forward's run reduction on the Zipf leg, the FM rank-ordered world-8 test
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_parity.py -x -q 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q -k "several_workers or compile_from_device or owner or windows" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_world8_fullsize.py -x -q -k "rank_ordered" 2>&1 | tail -15
line() {
python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.4f ms" % d["ms_per_step"], d.get("ms_per_step_repeats") and "median %.4f" % d["ms_per_step_repeats"]["median"],
          {k: round(v * 1e3, 1) for k, v in d["kernels_ms"].items() if v}, "wkb", d.get("ms_per_step_with_key_build"), "frac", round(d["roofline"]["frac"], 3))
except Exception as e:
    print(f, "FAILED", e)
PY
}
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --signal-keys 0 --force-sharded --general-path --schedule owner --no-cpu-baseline"
timeout 600 python bench.py --no-cpu-baseline --no-fm-leg > $O/bench_n1.json 2> $O/bench_n1.err; line $O/bench_n1.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c4/bench_n1.json").read().strip().splitlines()[-1])
z = d.get("zipf", {})
print("zipf", {k: z.get(k) for k in ("ms_per_step", "kernels_ms", "with_key_build_ms_per_step", "error")}, z.get("roofline", {}).get("frac"))
for t in d.get("table_sweep", {}).get("tables", []):
    print("sweep", {k: t.get(k) for k in ("keys_per_gpu", "ms_per_step", "kernels_ms", "with_key_build_ms_per_step", "error")}, t.get("roofline", {}).get("frac"))
PY
tail -3 $O/bench_n1.err
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/wkb -- \
    python $R/bench.py $N8 --steps 4 --warmup 2 --repeats 0 --batches 2 --no-owner-leg --key-build-steps 16 > $R/$O/wkb.json 2> $R/$O/wkb.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/sweep -- \
    python $R/bench.py --no-cpu-baseline --no-fm-leg --no-zipf-leg --steps 4 --warmup 2 --repeats 0 --batches 2 --key-build-steps 0 --sweep-keys 100000000 > $R/$O/sweep.json 2> $R/$O/sweep.err
cd $R
line $O/wkb.json
python - <<'PY'
import csv, glob
for d in ("wkb", "sweep"):
    for f in glob.glob("gpurun_out/r5c4/%s/**/*kernel_stats.csv" % d, recursive=True)[:1]:
        rows = [r for r in csv.DictReader(open(f)) if "k_kb" in r["Name"] or "k_own" in r["Name"] or "k_lr" in r["Name"] or "k_rows" in r["Name"]]
        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
        print(d)
        for r in rows[:16]:
            print("  %-70s calls %6s avg_us %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
