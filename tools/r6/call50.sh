#!/bin/bash
# round 6, call 50: hot keys get ranges of their own (a sorted sample ahead of the partition):
# the sort's tests, the worker-side build's test, the sort alone, the sequential cycle on the
# uniform and on a Zipf(1.1) stream (by hand / round 5's build)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 900 python -m pytest tests/test_gpu_keybuild.py -m gpu -x -q -k "sort_key_pos" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -k "worker_side_lr_build or world_one or device_arrays" 2>&1 | tail -5
timeout 600 python tools/r6/sort_probe.py 20
SEQ="--force-sharded --general-path --schedule sequential --no-cpu-baseline --steps 4 --warmup 2 --repeats 0 --batches 4 --no-owner-leg --key-build-steps 16"
for z in "" "--zipf 1.1 --signal-keys 0"; do
for t in 0 1; do
  timeout 600 python bench.py $SEQ $z --tune key_build=$t 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d.get('with_key_build_sharded') or d.get('with_key_build') or {}
print('$z key_build=$t', {k: (round(v,3) if isinstance(v,float) else v) for k,v in w.items() if k!='what'})"
done; done
