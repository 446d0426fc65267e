#!/bin/bash
# round-4 GPU call 1: gradient-kernel variants (correctness, timing, SQ counters), then the
# 8-rank full-size parity runs (first at 1/10 size)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c1
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_cells.py -x -q > $O/cells_tests.log 2>&1; echo "cells tests rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/cells_knobs.py --knobs 399,300,301,302,304,305,306,399,301 --steps 32 > $O/knobs.log 2>&1; echo "knobs rc=$?" | tee -a $O/summary.txt
cat $O/knobs.log | tee -a $O/summary.txt
timeout 300 python tools/cells_knobs.py --knobs 399,300,301,305 --steps 32 --zipf 1.1 > $O/knobs_zipf.log 2>&1
cat $O/knobs_zipf.log | tee -a $O/summary.txt
# SQ counters of the general kernel and of the compacting variant
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_g$i -- \
      python $R/tools/cells_knobs.py --knobs 399,301 --steps 8 > $O/pmc_g$i.log 2> $O/pmc_g$i.err
done
cd $R
python tools/pmc_summary.py $O/pmc_g 2>/dev/null | head -0
python - <<'PY' | tee -a $O/summary.txt
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/c1"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/pmc_g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "grad" not in k and "fwd" not in k: continue
        acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in acc.items():
    print("==",k)
    for c,vals in sorted(v.items()):
        vals.sort(); print("   %-24s %14.0f  (n=%d)"%(c, vals[len(vals)//2], len(vals)))
PY
# the 8-rank parity runs: small first, then full size
XF_WORLD8_ROWS=5000 timeout 600 python -m pytest tests/test_gpu_world8_fullsize.py -x -q > $O/world8_small.log 2>&1; rc=$?; echo "world8 small rc=$rc" | tee -a $O/summary.txt; tail -30 $O/world8_small.log | tee -a $O/summary.txt
if [ $rc = 0 ]; then
  /usr/bin/time -v timeout 1500 python -m pytest tests/test_gpu_world8_fullsize.py -x -q --durations=5 > $O/world8_full.log 2>&1; echo "world8 full rc=$?" | tee -a $O/summary.txt; tail -40 $O/world8_full.log | tee -a $O/summary.txt
fi
free -g | tee -a $O/summary.txt; nproc | tee -a $O/summary.txt
