#!/bin/bash
# The round's evidence in one go (GPU box, repo root): the default bench line, the rocprofv3
# kernel-trace stats of the same command, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) with
# the in-run stream calibration.  Every rocprofv3 call has a hard time limit.
#   bash tools/profile_round.sh gpurun_out/round
set -u
OUT=$1
R=$PWD
mkdir -p $OUT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- \
    python $R/bench.py --no-cpu-baseline --key-build-steps 0 > $R/$OUT/stats.json 2> $R/$OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -- \
      python $R/bench.py --steps 6 --warmup 8 --no-cpu-baseline --key-build-steps 0 --pmc-calibrate \
      > $R/$OUT/pmc_$c.json 2> $R/$OUT/pmc_$c.err
done
cd $R
python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_traffic.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/step_kernel_stats.csv
cp $(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $OUT/pmc_fetch_counter_collection.csv
cp $(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/pmc_write_counter_collection.csv
cp $OUT/pmc_traffic.json profiles/pmc_traffic_latest.json 2>/dev/null
tail -c 1500 $OUT/bench_n1.json
head -12 $OUT/step_kernel_stats.csv | cut -c1-160
# the other workloads' bench lines (no rocprof): power-law LR, FM configs, the N>1 code path at N=1
python bench.py --zipf 1.1 --no-cpu-baseline > $OUT/bench_zipf11.json 2> $OUT/bench_zipf11.err
python bench.py --model fm --k 16 --optimizer sgd --no-cpu-baseline > $OUT/bench_fm16_sgd.json 2> $OUT/bench_fm16_sgd.err
python bench.py --model fm --k 16 --optimizer ftrl --no-cpu-baseline > $OUT/bench_fm16_ftrl.json 2> $OUT/bench_fm16_ftrl.err
python bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --no-cpu-baseline > $OUT/bench_fm64_ftrl_zipf11.json 2> $OUT/bench_fm64_ftrl_zipf11.err
python bench.py --force-sharded --general-path --schedule sequential --no-cpu-baseline > $OUT/bench_general_path_n1_sequential.json 2> $OUT/gp_seq.err
python bench.py --force-sharded --general-path --schedule stale1 --no-cpu-baseline > $OUT/bench_general_path_n1_stale1.json 2> $OUT/gp_st.err
python bench.py --force-sharded --no-cpu-baseline > $OUT/bench_sharded_n1.json 2> $OUT/sh1.err
for f in bench_zipf11 bench_fm16_sgd bench_fm16_ftrl bench_fm64_ftrl_zipf11 bench_general_path_n1_sequential bench_general_path_n1_stale1 bench_sharded_n1; do
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", "%.4g ex/s" % d["value"], "%.4f ms" % d["ms_per_step"], {k: round(v*1e3,1) for k,v in d["kernels_ms"].items() if v}, "8d %.0f GB/s" % d["step_gbs_survey_8d"])
except Exception as e:
    print("$f", "FAILED", e)
PY
done
