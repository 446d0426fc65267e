#!/bin/bash
# round 6, call 49: the scans by hand under the device key build (its tests, FM's), the driver
# line with exchange_worker_side, kernel stats of the sequential cycle and of the sort alone
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r06d; mkdir -p $OUT
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_keybuild.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06d/bench_n1.json").read().splitlines() if l.startswith("{")][-1])
print("value %.4g ms %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
print("exchange_worker_side", {k: v for k, v in d["n8_shape"].get("exchange_worker_side", {}).items() if k != "what"})
print("summary", json.dumps(d["summary"]))
PY
R=$PWD
stats() {
  local name=$1; shift
  (cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_$name -- "$@" > /tmp/_$name.out 2> /tmp/_$name.err)
  cp $(find /tmp/_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv 2>/dev/null
  grep "by hand" /tmp/_$name.out
  rm -rf /tmp/_$name /tmp/_$name.out /tmp/_$name.err
}
stats seq_worker_side python $R/bench.py --force-sharded --general-path --schedule sequential --no-cpu-baseline --steps 4 --warmup 2 --repeats 0 --batches 4 --no-owner-leg --key-build-steps 16
stats sort_key_pos python $R/tools/r6/sort_probe.py 10
head -12 $OUT/seq_worker_side_kernel_stats.csv | cut -c1-150
