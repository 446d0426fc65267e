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
