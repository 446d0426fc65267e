#!/bin/bash
# round 6, call 24: the HIP calls of a steady-state minibatch of the first-epoch leg (what is the
# 0.63 ms per minibatch made of when xf_lr_update_dev itself takes 0.40?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/_tl -- python $GRAFT_REPO_ROOT/tools/r6/fresh_probe.py 10000000 24 > /tmp/_tl.out 2>&1)
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
# the last two keyed builds: from the second-to-last k_kb_hist on
hs = [k for k in ks if "k_kb_hist" in k[2]]
t0 = hs[-3][0] - 100_000
t1 = hs[-1][0] - 100_000
win = [e for e in ev + ks if t0 <= e[0] <= t1]
win.sort()
print("two minibatches: %.1f us" % ((t1 - t0) / 1e3))
for s, e, n in win:
    if n.startswith("KERN") or e - s > 3000 or 60_000 < s - t0 < 270_000:
        print("%9.1f us  %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
