#!/bin/bash
# GPU box: key-build tests, then rocprofv3 kernel stats of the key-build loop (tools/kb_knobs.py)
#   bash tools/kb_prof.sh <outdir-under-gpurun_out> [extra kb_knobs args]
set -u
OUT=gpurun_out/$1; shift
R=$PWD
mkdir -p $OUT
python -m pytest tests/test_gpu_keybuild.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -- \
    python $R/tools/kb_knobs.py --knobs 0 --iters 20 "$@" > $R/$OUT/knobs_prof.txt 2>&1
cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    n=r["Name"]
    if any(k in n for k in ("k_kb","k_plan","k_items","copyBuffer","fillBuffer","k_lr","k_lookup","k_resolve","k_cell","rocprim")):
        i=n.find("k_"); nm=n[i:i+34] if i>=0 else n[:34]
        print("%-36s calls %4s avg %9.1f us  min %9.1f" % (nm, r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
tail -2 $OUT/knobs_prof.txt
