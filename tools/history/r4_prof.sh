#!/bin/bash
# rocprofv3 kernel stats of one python command (GPU box): tools/r4_prof.sh <tag> <python args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python "$@" > $O/out.txt 2>&1
tail -4 $O/out.txt
cd $R
python - <<PY
import csv,glob
f=glob.glob("$O/prof/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:24]:
    n=r["Name"]; i=n.find("k_"); nm=n[i:i+60] if i>=0 else n[:60]
    if "rocprim" in n: nm="rocprim:"+("onesweep" if "onesweep" in n else "scan" if "scan" in n else n[40:80])
    print("%-60s calls %4s avg %9.1f us total %8.1f ms" % (nm[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
