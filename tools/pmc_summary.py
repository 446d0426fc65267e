#!/usr/bin/env python3
"""Median per-kernel value of every counter found under <dir>/g*/ (steady-state launches)."""
import csv, glob, os, re, sys, collections
def kname(full):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", full)
    return (m.group(1) + (m.group(2) or "")) if m else full.split("(")[0].strip()
d = sys.argv[1]
tab = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(d, "g*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        if k.startswith("k_") and not k.startswith("k_calib") and k != "k_fill_u64":
            per[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            per[(k, "dur_us")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (k, c), v in per.items():
        v = v[len(v) // 2:]
        tab[k][c] = sorted(v)[len(v) // 2]
for k in sorted(tab):
    print("==", k)
    for c in sorted(tab[k]):
        print("   %-40s %16.1f" % (c, tab[k][c]))
