#!/bin/bash
# round 5, GPU call 2: the whole GPU suite on the new paths, where the N = 8 compile + step goes,
# the several-workers pass at 1024 threads, the resolve with per-lane LDS atomics, the division
# guard gone
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c2
mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -25
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --signal-keys 0 --force-sharded --general-path --schedule owner --no-cpu-baseline"
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
for K in 0 296; do
XF_OWNER_TIMING_SOURCES=8 timeout 400 python bench.py $N8 --repeats 3 --batches 8 --no-owner-leg --key-build-steps 0 --exp-knob $K > $O/n8_src8_k$K.json 2> $O/n8_src8_k$K.err; line $O/n8_src8_k$K.json
done
timeout 300 python bench.py --no-cpu-baseline --no-fm-leg --no-zipf-leg --no-table-sweep --repeats 5 > $O/bench_lr.json 2> $O/bench_lr.err; line $O/bench_lr.json
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $R/$O/wkb -- \
    python $R/bench.py $N8 --steps 4 --warmup 2 --repeats 0 --batches 2 --no-owner-leg --key-build-steps 16 > $R/$O/wkb.json 2> $R/$O/wkb.err
cd $R
line $O/wkb.json
python - <<'PY'
import csv, glob
for pat, n in (("gpurun_out/r5c2/wkb/**/*kernel_stats.csv", 22), ("gpurun_out/r5c2/wkb/**/*hip_api_stats.csv", 14)):
    for f in glob.glob(pat, recursive=True)[:1]:
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
        print(f.split("/")[-1])
        for r in rows[:n]:
            print("  %-70s calls %6s avg_us %9.1f total_ms %8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
for K in 0 116; do
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kb$K -- \
    python $R/tools/kb_knobs.py --knobs $K --iters 16 --step > $R/$O/kb$K.txt 2>&1
cd $R
tail -1 $O/kb$K.txt
python - "$K" <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/r5c2/kb%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(("k_kb", "k_lr", "void (anonymous namespace)::k_kb", "void (anonymous namespace)::k_lr")) or "k_kb_" in r["Name"] or "k_lr_" in r["Name"]:
            print("  %-60s calls %5s avg_us %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
rm -rf $O/wkb/*/*.db $O/kb0 $O/kb116 2>/dev/null
find $O -name "*.csv" -size +2M -delete
