#!/bin/bash
# Round-6 evidence in one go (GPU box, repo root): bench lines, rocprofv3 kernel stats, and the
# size-resolved PMC traffic (tools/pmc2.sh) of the workloads a number is quoted for — this round
# also the Zipf(1.1) step, the first-epoch leg and the N = 8 owner shape with its hot field.
#   bash tools/profile_round6.sh gpurun_out/r06 [nopmc]
set -u
OUT=$1; MODE=${2:-}
R=$PWD
mkdir -p $OUT
HEAD=$(cat .git_head 2>/dev/null || git rev-parse --short HEAD 2>/dev/null || echo unknown)
line() {  # one line per bench JSON
  python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1])
    print(f.split("/")[-1], "%.4g ex/s" % d["value"], "%.4f ms" % d["ms_per_step"],
          d.get("ms_per_step_repeats") and "median %.4f" % d["ms_per_step_repeats"]["median"],
          {k: round(v * 1e3, 1) for k, v in d["kernels_ms"].items() if v},
          "wkb", d.get("ms_per_step_with_key_build"), "frac", round(d["roofline"]["frac"], 3),
          "logloss", d.get("logloss", {}).get("natural"))
    if "summary" in d:
        print("   summary", json.dumps(d["summary"]))
except Exception as e:
    print(f, "FAILED", e)
PY
}
stats() {  # stats <name> <command...>: rocprofv3 --kernel-trace --stats, the summary kept
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv \
      -d /tmp/_$name -- "$@" > /tmp/_$name.out 2> /tmp/_$name.err)
  cp $(find /tmp/_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv 2>/dev/null
  rm -rf /tmp/_$name /tmp/_$name.out /tmp/_$name.err
}
QUIET="--no-cpu-baseline --no-fresh-table --no-n8-shape --no-end-to-end --sustained-seconds 0"
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000"
N8O="$N8 --force-sharded --general-path --schedule owner --no-cpu-baseline"
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; line $OUT/bench_n1.json
stats step python $R/bench.py $QUIET --key-build-steps 0 --repeats 0 --no-fm-leg --no-zipf-leg --no-table-sweep
stats key_build python $R/tools/kb_knobs.py --knobs 0 --iters 24 --step
stats fresh_table_1e7 python $R/tools/r6/fresh_probe.py 10000000 40
stats fresh_table_1e8 python $R/tools/r6/fresh_probe.py 100000000 40
stats n8_key_build python $R/bench.py $N8O --signal-keys 0 --steps 4 --warmup 2 --repeats 0 --batches 2 --no-owner-leg --key-build-steps 16
# the worker side of the weight / gradient exchange (schedule sequential, one rank): compile from
# device arrays + step per minibatch; the hand-written sort alone on the three streams
stats seq_worker_side python $R/bench.py --force-sharded --general-path --schedule sequential --no-cpu-baseline --steps 4 --warmup 2 --repeats 0 --batches 4 --no-owner-leg --key-build-steps 16
stats sort_key_pos python $R/tools/r6/sort_probe.py
stats sweep_1e8 python $R/bench.py $QUIET --no-fm-leg --no-zipf-leg --steps 4 --warmup 2 --repeats 0 --batches 2 --key-build-steps 0 --sweep-keys 100000000
stats zipf11 python $R/bench.py $QUIET --zipf 1.1 --signal-keys 0 --batches 8 --key-build-steps 0 --repeats 0 --no-fm-leg
stats fm python $R/tools/fm_leg.py --batches 4
python bench.py --zipf 1.1 --no-cpu-baseline > $OUT/bench_zipf11.json 2> $OUT/bench_zipf11.err; line $OUT/bench_zipf11.json
python bench.py --model fm --k 16 --optimizer sgd --no-cpu-baseline --repeats 3 --batches 8 > $OUT/bench_fm16_sgd.json 2> $OUT/bench_fm16_sgd.err; line $OUT/bench_fm16_sgd.json
python bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --no-cpu-baseline --repeats 3 --batches 8 > $OUT/bench_fm64_ftrl_zipf11.json 2> $OUT/bench_fm64_ftrl_zipf11.err; line $OUT/bench_fm64_ftrl_zipf11.json
[ "$MODE" = "nopmc" ] && exit 0
# memory-side traffic, request sizes resolved, calibration patterns in the same runs
B="--steps 6 --warmup 8 $QUIET --key-build-steps 0 --repeats 0 --no-fm-leg --no-zipf-leg --batches 8 --pmc-calibrate"
bash tools/pmc2.sh $OUT lr python $R/bench.py $B --no-table-sweep 2>&1 | grep -v "^  k_\(build\|fill\|id_wr\|list\|move\|rehash\|take\|count\|cell\|blk\|df\|ar\)"
bash tools/pmc2.sh $OUT lr_zipf11 python $R/bench.py $B --zipf 1.1 --signal-keys 0 --no-table-sweep 2>&1 | grep "k_lr\|pmc2"
bash tools/pmc2.sh $OUT sweep_1e8 python $R/bench.py $B --batches 2 --sweep-keys 100000000 2>&1 | grep "k_lr\|pmc2"
bash tools/pmc2.sh $OUT key_build python $R/tools/kb_knobs.py --knobs 0 --iters 8 --step --pmc-calibrate 2>&1 | grep "k_kb\|k_lr\|pmc2"
XF_OWNER_TIMING_SOURCES=8 bash tools/pmc2.sh $OUT n8_shard_shape_owner_8_sources python $R/bench.py $N8O --signal-keys 0 --no-owner-leg $B --no-table-sweep 2>&1 | grep "k_lr\|k_owner\|k_sum\|pmc2"
bash tools/pmc2.sh $OUT n8_shard_shape_owner_sum_then_step python $R/bench.py $N8O --signal-keys 0 --no-owner-leg $B --no-table-sweep 2>&1 | grep "k_lr\|k_owner\|k_sum\|pmc2"
# the vector-memory path of the N = 8 owner shape's kernels (TA busy, L1 -> L2 requests and their
# latency, address translation): why the forward at 32 windows and the several-workers pass sit
# where they do
XF_OWNER_TIMING_SOURCES=8 bash tools/pmc_mem_passes.sh $OUT/_mem_n8 $N8O --signal-keys 0 --no-owner-leg --repeats 0 --batches 4 > $OUT/pmc_mem_n8_owner_8_sources.txt 2>&1
bash tools/pmc_mem_passes.sh $OUT/_mem_n1 $QUIET --repeats 0 --batches 8 --no-fm-leg --no-zipf-leg --no-table-sweep > $OUT/pmc_mem_lr.txt 2>&1
rm -rf $OUT/_mem_n8 $OUT/_mem_n1
rm -f $OUT/*_rd.json $OUT/*_wr.json
for f in $OUT/pmc_traffic_*.json; do python - "$f" "$HEAD" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); d["git_head"] = sys.argv[2]; json.dump(d, open(sys.argv[1], "w"), indent=1)
PY
done
# end to end through xflow_lr on a 3 GB file (text, GPU tokeniser, block cache)
python tools/e2e_text.py 1200000 $OUT/e2e.json > $OUT/e2e.log 2>&1; tail -1 $OUT/e2e.log | cut -c1-600
