#!/bin/bash
# Experiment (GPU box): SQ counters of the FM step's kernels (one rocprofv3 --pmc pass with the
# kernel trace only), table-resident records on and off.  Output: gpurun_out/pmc_fm_{rec,norec}.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in rec norec; do
  if [ $mode = norec ]; then export XF_FM_TABLE_RECORDS=0; else unset XF_FM_TABLE_RECORDS; fi
  rm -rf /tmp/pmcfm_$mode
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
      SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
      -d /tmp/pmcfm_$mode -o p --output-format csv -- python $R/tools/fm_leg.py --batches 4 > /tmp/pmcfm_$mode.log 2>&1
  f=$(find /tmp/pmcfm_$mode -name "*counter_collection.csv" | head -1)
  python3 - "$f" $mode <<'PY'
import csv, re, sys, collections
f, mode = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    m = re.search(r"k_fm_\w+(<[^>]*>)?", r["Kernel_Name"])
    if not m: continue
    k = m.group(0)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k in acc:
    print(mode, k, "launches", n[k], {c: round(v / max(n[k], 1)) for c, v in acc[k].items()})
PY
done
