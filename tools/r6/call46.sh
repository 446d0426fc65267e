#!/bin/bash
# round 6, call 46: the worker side of the weight / gradient exchange (schedule sequential, the
# worker's default): xf_sharded_compile_dev + the step per minibatch with the hand-written sort
# and with the library's (key_build = 1), and the kernels of one cycle
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
SEQ="--force-sharded --general-path --schedule sequential --no-cpu-baseline --steps 4 --warmup 2 --repeats 0 --batches 4 --no-owner-leg --key-build-steps 16"
for t in 0 1; do
  timeout 600 python bench.py $SEQ --tune key_build=$t 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d.get('with_key_build_sharded') or d.get('with_key_build') or {}
print('key_build=$t', {k: (round(v,3) if isinstance(v,float) else v) for k,v in w.items() if k!='what'})"
done
rm -rf /tmp/_p
(cd /tmp && PYTHONPATH=$GRAFT_REPO_ROOT timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_p -- python $GRAFT_REPO_ROOT/bench.py $SEQ > /tmp/_p.out 2>&1)
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/_p/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:28]:
        print("%-70s calls %4s avg %8.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
