#!/bin/bash
# round 6, call 41: the N = 8 owner shape's compile + step cycle: kernels and HIP calls of one cycle
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
rm -rf /tmp/_tl
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --force-sharded --general-path --schedule owner --no-cpu-baseline"
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/_tl -- python $GRAFT_REPO_ROOT/bench.py $N8 --signal-keys 0 --steps 4 --warmup 2 --repeats 0 --batches 2 --no-owner-leg --key-build-steps 16 > /tmp/_tl.out 2>&1)
python - <<'PY'
import csv, glob, re
d = "/tmp/_tl"
ev = []
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api  " + r["Function"]))
ks = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERN " + (m.group(0) if m else r["Kernel_Name"])[:50]))
ks.sort()
hs = [k for k in ks if "k_rows_of_nnz" in k[2]]
t0, t1 = hs[-2][0] - 30_000, hs[-1][0] - 30_000
print("one cycle: %.1f us" % ((t1 - t0) / 1e3))
win = sorted(e for e in ev + ks if t0 <= e[0] <= t1)
for s, e, n in win:
    if n.startswith("KERN") or e - s > 6000:
        print("%9.1f us  %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
