R=$PWD; OUT=gpurun_out/pmc_shape; mkdir -p $OUT
ARGS="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --steps 6 --warmup 4 --no-cpu-baseline --key-build-steps 0 --pmc-calibrate"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -- python $R/bench.py $ARGS > $R/$OUT/pmc_$c.json 2> $R/$OUT/pmc_$c.err
done
cd $R
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_traffic.json
python - <<PY
import json
d=json.load(open("$OUT/pmc_traffic.json"))
print(d["workload"])
for k,e in d["kernels"].items():
    if "lr_" in k: print("%-30s launches %3d traffic %8.1f MB  %7.1f us" % (k[:30], e["launches"], e["traffic"]/1e6, e["median_us_under_pmc"]))
PY
