#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c5
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_cells.py -x -q -k variant > $O/t1.log 2>&1; echo "variants rc=$?" | tee -a $O/summary.txt; tail -3 $O/t1.log | tee -a $O/summary.txt
timeout 300 python tools/cells_knobs.py --knobs 300,428,304,432,300,428 --steps 32 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
cd /tmp
i=0
for grp in "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_g$i -- \
      python $R/tools/cells_knobs.py --knobs 300,316,332 --steps 8 > $O/pmc_g$i.log 2> $O/pmc_g$i.err
  echo "pmc group $i rc=$?" | tee -a $O/summary.txt
done
cd $R
python - <<'PY' | tee -a $O/summary.txt
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/c5"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/pmc_g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "grad_dense" not in k and "fwd_cells" not in k: continue
        k=k.replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","")[:34]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names=sorted({c for v in acc.values() for c in v})
ks=sorted(acc)
print("%-40s"%"counter"+"".join("%22s"%k[:21] for k in ks))
for c in names:
    def med(v):
        v=sorted(v); return v[len(v)//2] if v else float('nan')
    print("%-40s"%c+"".join("%22.0f"%med(acc[k].get(c,[])) for k in ks))
PY
