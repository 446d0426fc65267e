#!/bin/bash
# round 6, call 6: FM on the overlapped owner schedule + the defrag's larger blocks under the
# suites; the first minibatches of an empty table as a kernel timeline; the whole bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log   # (no-op when the library matches the sources)
timeout 2000 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_keybuild.py tests/test_gpu_cli.py \
  tests/test_gpu_ingest.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r6/call6_tests.log
cat gpurun_out/r6/call6_tests.log
rm -rf /tmp/ft
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -- \
    python "$GRAFT_REPO_ROOT/tools/r6/fresh_probe.py" 10000000 3 30 > /tmp/ft.out 2> /tmp/ft.err)
tail -1 /tmp/ft.out | cut -c1-400
python tools/r6/timeline.py /tmp/ft 0 110
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6/call6_bench.json 2> gpurun_out/r6/call6_bench.err ) 2>&1 | tail -4
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6/call6_bench.json").read().splitlines() if l.startswith("{")][-1])
print(json.dumps(d["summary"]))
print("zipf", d["zipf"]["ms_per_step"], d["zipf"]["kernels_ms"])
PY
tail -3 gpurun_out/r6/call6_bench.err
