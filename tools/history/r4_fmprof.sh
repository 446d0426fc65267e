#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/tools/fm_kb_prof.py "$@" > $O/out.txt 2>&1
grep "ms per" $O/out.txt || tail -20 $O/out.txt
cd $R
python - <<PY
import csv,glob
f=glob.glob("$O/prof/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:30]:
    n=r["Name"]; i=n.find("k_"); nm=n[i:i+40] if i>=0 else n[:60]
    if "rocprim" in n: nm="rocprim:"+("onesweep" if "onesweep" in n else "scan" if "scan" in n else n[40:80])+(" u64" if "unsigned long" in n[:300] else "")
    print("%-50s calls %4s avg %9.1f us total %8.1f ms" % (nm[:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
