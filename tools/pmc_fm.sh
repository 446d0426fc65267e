R=$PWD; OUT=gpurun_out/pmc_fm; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/a -- python $R/bench.py --model fm --k 16 --optimizer sgd --steps 4 --warmup 3 --no-cpu-baseline > $R/$OUT/a.json 2> $R/$OUT/a.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/$OUT/b -- python $R/bench.py --model fm --k 16 --optimizer sgd --steps 4 --warmup 3 --no-cpu-baseline > $R/$OUT/b.json 2> $R/$OUT/b.err
cd $R
python - <<'PY'
import csv,glob,collections
for d in ("a","b"):
    fs=glob.glob("gpurun_out/pmc_fm/%s/**/*counter_collection.csv"%d, recursive=True)
    if not fs:
        print(d,"no csv", open("gpurun_out/pmc_fm/%s.err"%d).read()[-600:]); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"][:50]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in agg.items():
        if "fm_grad_tiled" in k or "fm_forward_scalars" in k or "gather_scalars" in k:
            print(k, {c:"%.3g"%x for c,x in v.items()})
PY
