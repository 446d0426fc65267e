#!/bin/bash
# Round-3 evidence in one go (GPU box, repo root): the default bench line, rocprofv3 kernel
# stats of the step and of the key-build loop, bench lines of the other workloads.
#   bash tools/profile_round3.sh gpurun_out/r03
set -u
OUT=$1
R=$PWD
mkdir -p $OUT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- \
    python $R/bench.py --no-cpu-baseline --key-build-steps 0 --repeats 0 --no-fm-leg > $R/$OUT/stats.json 2> $R/$OUT/stats.err
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kbstats -- \
    python $R/tools/kb_knobs.py --knobs 0 --iters 24 --step > $R/$OUT/kbstats.txt 2>&1
cd $R
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/step_kernel_stats.csv
cp $(find $OUT/kbstats -name "*kernel_stats.csv" | head -1) $OUT/key_build_kernel_stats.csv
python bench.py --zipf 1.1 --no-cpu-baseline > $OUT/bench_zipf11.json 2> $OUT/bench_zipf11.err
python bench.py --model fm --k 16 --optimizer sgd --no-cpu-baseline --repeats 3 > $OUT/bench_fm16_sgd.json 2> $OUT/bench_fm16_sgd.err
python bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --no-cpu-baseline --repeats 3 > $OUT/bench_fm64_ftrl_zipf11.json 2> $OUT/bench_fm64_ftrl_zipf11.err
python bench.py --rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --force-sharded --general-path --schedule owner --no-cpu-baseline --repeats 3 > $OUT/bench_n8_shard_shape_owner.json 2> $OUT/n8.err
python bench.py --rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --no-cpu-baseline --repeats 3 > $OUT/bench_n8_shard_shape_fused.json 2> $OUT/n8f.err
for f in bench_n1 bench_zipf11 bench_fm16_sgd bench_fm64_ftrl_zipf11 bench_n8_shard_shape_owner bench_n8_shard_shape_fused; do
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", "%.4g ex/s" % d["value"], "%.4f ms" % d["ms_per_step"], d.get("ms_per_step_repeats") and "median %.4f" % d["ms_per_step_repeats"]["median"], {k: round(v*1e3,1) for k,v in d["kernels_ms"].items() if v}, "wkb", d.get("ms_per_step_with_key_build"))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
rm -rf $OUT/stats $OUT/kbstats
# FM: the fm leg's kernels (rocprofv3 stats), its HBM traffic (two PMC passes), the FM step through
# the owner-compute exchange path at world 1, configs[4]'s shard shape
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/fmstats -- \
    python $R/tools/fm_leg.py --batches 4 > $R/$OUT/fm_leg.json 2> $R/$OUT/fm_leg.err
cd $R
cp $(find $OUT/fmstats -name "*kernel_stats.csv" | head -1) $OUT/fm_kernel_stats.csv
rm -rf $OUT/fmstats
FMLEG=1 bash tools/pmc_fm_traffic.sh $OUT/pmc_fm 2>&1 | tail -6
cp $OUT/pmc_fm/pmc_traffic.json $OUT/pmc_traffic_fm16_sgd.json
rm -rf $OUT/pmc_fm
python bench.py --force-sharded --general-path --model fm --k 16 --optimizer sgd --schedule owner --batches 4 --no-cpu-baseline --repeats 3 > $OUT/bench_fm16_sgd_owner_exchange_path.json 2> $OUT/fmo.err
python bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --keys-per-gpu 125000000 --capacity 64000000 --no-cpu-baseline --repeats 3 > $OUT/bench_cfg4_shard_shape_fm64_ftrl_zipf11_125Mkeys.json 2> $OUT/cfg4.err
for f in bench_fm16_sgd_owner_exchange_path bench_cfg4_shard_shape_fm64_ftrl_zipf11_125Mkeys; do
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", "%.4g ex/s" % d["value"], "%.4f ms" % d["ms_per_step"], {k: round(v*1e3,1) for k,v in d["kernels_ms"].items() if v}, d["config"].get("table_keys_touched"))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
# the step's PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, in-run stream calibration)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -- \
      python $R/bench.py --steps 6 --warmup 8 --no-cpu-baseline --key-build-steps 0 --repeats 0 --no-fm-leg --pmc-calibrate \
      > $R/$OUT/pmc_$c.json 2> $R/$OUT/pmc_$c.err
done
cd $R
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_traffic.json
cp $(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $OUT/pmc_fetch_counter_collection.csv
cp $(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/pmc_write_counter_collection.csv
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
python - <<PY
import json
d=json.load(open("$OUT/pmc_traffic.json"))
print(d["workload"]); print(d["calibration_true_bytes_per_counted_byte"])
for k,e in d["kernels"].items():
    if "k_lr" in k:
        print("%-40s fetch %7.1f MB  write %7.1f MB  traffic %7.1f MB  %6.1f us" % (k[:40], e["fetch_corrected"]/1e6, e["write_corrected"]/1e6, e["traffic"]/1e6, e["median_us_under_pmc"]))
PY
bash tools/pmc_keybuild.sh $OUT 2>&1 | tail -9
