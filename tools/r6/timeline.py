#!/usr/bin/env python3
"""Print a rocprofv3 kernel trace (csv) as a timeline: start offset, duration, gap to the previous
kernel's end, name.   python tools/r6/timeline.py <dir> [first] [count]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 120
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
# start the listing at the last k_div_exact (table creation of the probe's real table) if present
idx = [i for i, r in enumerate(rows) if "k_div_exact" in r["Kernel_Name"]]
if idx and not first:
    first = idx[-1]
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    m = re.search(r"k_\w+(<[^>]*>)?", name)
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%10.1f us  dur %8.1f  gap %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap,
                                                  (m.group(0) if m else name)[:60]))
    prev_end = e
