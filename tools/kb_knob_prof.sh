#!/bin/bash
# GPU box: per-kernel times of the key build under a list of exp_knob values (one rocprofv3 run each)
#   bash tools/kb_knob_prof.sh "0 104 108" [extra kb_knobs args]
R=$PWD
KN=$1; shift
cd /tmp && export TMPDIR=/tmp
for k in $KN; do
  rm -rf /tmp/kbk_$k
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kbk_$k -- \
      python $R/tools/kb_knobs.py --knobs $k --iters 10 "$@" > /tmp/kbk_$k.txt 2>&1
  f=$(find /tmp/kbk_$k -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv
out=[]
for r in csv.DictReader(open("$f")):
    n=r["Name"]
    if "k_kb" in n or "k_plan" in n:
        i=n.find("k_"); nm=n[i:i+14]
        out.append("%s %.1f" % (nm, float(r["AverageNs"])/1e3))
print("knob $k:", "  ".join(sorted(out)))
PY
done
