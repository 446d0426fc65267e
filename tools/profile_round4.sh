#!/bin/bash
# Round-4 evidence in one go (GPU box, repo root): bench lines, rocprofv3 kernel stats, and the
# size-resolved PMC traffic (tools/pmc2.sh) of every workload a number is quoted for.
#   bash tools/profile_round4.sh gpurun_out/r04 [quick]
set -u
OUT=$1; QUICK=${2:-}
R=$PWD
mkdir -p $OUT
line() {  # one line per bench JSON
  python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.4g ex/s" % d["value"], "%.4f ms" % d["ms_per_step"],
          d.get("ms_per_step_repeats") and "median %.4f" % d["ms_per_step_repeats"]["median"],
          {k: round(v * 1e3, 1) for k, v in d["kernels_ms"].items() if v},
          "wkb", d.get("ms_per_step_with_key_build"), "frac", round(d["roofline"]["frac"], 3),
          "logloss", d.get("logloss", {}).get("natural"))
except Exception as e:
    print(f, "FAILED", e)
PY
}
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; line $OUT/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- \
    python $R/bench.py --no-cpu-baseline --key-build-steps 0 --repeats 0 --no-fm-leg > $R/$OUT/stats.json 2> $R/$OUT/stats.err
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kbstats -- \
    python $R/tools/kb_knobs.py --knobs 0 --iters 24 --step > $R/$OUT/kbstats.txt 2>&1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/fmstats -- \
    python $R/tools/fm_leg.py --batches 4 > $R/$OUT/fm_leg.json 2> $R/$OUT/fm_leg.err
cd $R
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/step_kernel_stats.csv
cp $(find $OUT/kbstats -name "*kernel_stats.csv" | head -1) $OUT/key_build_kernel_stats.csv
cp $(find $OUT/fmstats -name "*kernel_stats.csv" | head -1) $OUT/fm_kernel_stats.csv
rm -rf $OUT/stats $OUT/kbstats $OUT/fmstats
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --signal-keys 0"
python bench.py --zipf 1.1 --no-cpu-baseline > $OUT/bench_zipf11.json 2> $OUT/bench_zipf11.err; line $OUT/bench_zipf11.json
python bench.py --model fm --k 16 --optimizer sgd --no-cpu-baseline --repeats 3 --batches 8 > $OUT/bench_fm16_sgd.json 2> $OUT/bench_fm16_sgd.err; line $OUT/bench_fm16_sgd.json
python bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --no-cpu-baseline --repeats 3 --batches 8 > $OUT/bench_fm64_ftrl_zipf11.json 2> $OUT/bench_fm64_ftrl_zipf11.err; line $OUT/bench_fm64_ftrl_zipf11.json
python bench.py $N8 --force-sharded --general-path --schedule owner --no-cpu-baseline --repeats 3 --batches 8 > $OUT/bench_n8_shard_shape_owner.json 2> $OUT/n8.err; line $OUT/bench_n8_shard_shape_owner.json
XF_OWNER_TIMING_SOURCES=8 python bench.py $N8 --force-sharded --general-path --schedule owner --no-cpu-baseline --repeats 3 --batches 8 --no-owner-leg > $OUT/bench_n8_shard_shape_owner_8_pretended_sources.json 2> $OUT/n8s.err; line $OUT/bench_n8_shard_shape_owner_8_pretended_sources.json
python bench.py $N8 --no-cpu-baseline --repeats 3 --batches 8 > $OUT/bench_n8_shard_shape_fused.json 2> $OUT/n8f.err; line $OUT/bench_n8_shard_shape_fused.json
if [ -z "$QUICK" ] || [ "$QUICK" = "benchonly" ]; then
python bench.py --force-sharded --general-path --model fm --k 16 --optimizer sgd --schedule owner --batches 4 --no-cpu-baseline --repeats 3 > $OUT/bench_fm16_sgd_owner_exchange_path.json 2> $OUT/fmo.err; line $OUT/bench_fm16_sgd_owner_exchange_path.json
python bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --keys-per-gpu 125000000 --capacity 64000000 --no-cpu-baseline --repeats 3 --batches 8 > $OUT/bench_cfg4_shard_shape_fm64_ftrl_zipf11_125Mkeys.json 2> $OUT/cfg4.err; line $OUT/bench_cfg4_shard_shape_fm64_ftrl_zipf11_125Mkeys.json
fi
[ "$QUICK" = "benchonly" ] && exit 0   # (a second box for the timings: the traffic is the box's own)
# memory-side traffic, request sizes resolved, calibration patterns in the same runs
B="--steps 6 --warmup 8 --no-cpu-baseline --key-build-steps 0 --repeats 0 --no-fm-leg --batches 8 --pmc-calibrate"
bash tools/pmc2.sh $OUT lr python $R/bench.py $B 2>&1 | grep -v "^  k_\(build\|fill\|id_wr\|list\|move\|rehash\|take\|count\|cell\|blk\)"
bash tools/pmc2.sh $OUT lr_zipf11 python $R/bench.py --zipf 1.1 $B 2>&1 | grep "k_lr\|pmc2"
bash tools/pmc2.sh $OUT n8_shard_shape_owner python $R/bench.py $N8 --force-sharded --general-path --schedule owner --no-owner-leg $B 2>&1 | grep "k_lr\|k_owner\|k_sum\|pmc2"
bash tools/pmc2.sh $OUT fm16_sgd python $R/tools/fm_leg.py --batches 4 --pmc-calibrate 2>&1 | grep "k_fm\|pmc2"
bash tools/pmc2.sh $OUT fm64_ftrl_zipf11 python $R/bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --steps 6 --warmup 4 --no-cpu-baseline --repeats 0 --batches 8 --pmc-calibrate 2>&1 | grep "k_fm\|pmc2"
bash tools/pmc2.sh $OUT key_build python $R/tools/kb_knobs.py --knobs 0 --iters 8 --step --pmc-calibrate 2>&1 | grep "k_kb\|k_lr\|pmc2"
rm -f $OUT/*_rd.json $OUT/*_wr.json
