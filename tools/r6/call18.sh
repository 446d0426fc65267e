#!/bin/bash
# round 6, call 18: where the first minibatch's host time goes (kernel timeline with gaps + the
# HIP calls in between)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/_tl -- python $GRAFT_REPO_ROOT/tools/r6/fresh_probe.py 10000000 3 > /tmp/_tl.out 2>&1)
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
# the window: from the last k_div_exact to 3 ms later
t0 = [k for k in ks if "k_ar_ranges" in k[2]][1][0] - 100_000
win = [e for e in ev + ks if t0 <= e[0] <= t0 + 1_600_000]
win.sort()
for s, e, n in win:
    if n.startswith("KERN") or e - s > 4000:
        print("%9.1f us  %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
