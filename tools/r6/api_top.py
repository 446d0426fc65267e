#!/usr/bin/env python3
"""The longest HIP API calls and kernels of a rocprofv3 csv trace, in time order.
    python tools/r6/api_top.py <dir> [count]"""
import csv
import glob
import re
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
ev = []
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api  " + r["Function"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                   "kern " + (m.group(0) if m else r["Kernel_Name"])[:50]))
if not ev:
    raise SystemExit("no trace under " + d)
t0 = min(e[0] for e in ev)
top = sorted(ev, key=lambda e: e[0] - e[1])[:n]
for s, e, name in sorted(top):
    print("%12.1f us  dur %10.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name))
