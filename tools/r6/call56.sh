#!/bin/bash
# round 6, call 56: kernel stats of the sequential schedule's worker side (uniform and Zipf 1.1)
# and of the sort alone, the round's last library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r06f; mkdir -p $OUT
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
R=$PWD
stats() {
  local name=$1; shift
  (cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_$name -- "$@" > /tmp/_$name.out 2> /tmp/_$name.err)
  cp $(find /tmp/_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv 2>/dev/null
  grep "by hand" /tmp/_$name.out
  tail -1 /tmp/_$name.out | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); w=d.get('with_key_build_sharded') or {}
    print('$name', {k: round(v,3) for k,v in w.items() if isinstance(v,float)})
except Exception: pass"
  rm -rf /tmp/_$name /tmp/_$name.out /tmp/_$name.err
}
SEQ="--force-sharded --general-path --schedule sequential --no-cpu-baseline --steps 4 --warmup 2 --repeats 0 --batches 4 --no-owner-leg --key-build-steps 16"
stats seq_worker_side python $R/bench.py $SEQ
stats seq_worker_side_zipf11 python $R/bench.py $SEQ --zipf 1.1 --signal-keys 0
stats sort_key_pos python $R/tools/r6/sort_probe.py 10
for f in seq_worker_side seq_worker_side_zipf11 sort_key_pos; do echo == $f; head -16 $OUT/${f}_kernel_stats.csv | cut -d, -f1,2,4 | cut -c1-110; done
