#!/bin/bash
# round 5, GPU call 23: the library as committed at the end of the round — cells / key build / FM
# tests, smoke, and the driver's line once more
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c23
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_keybuild.py tests/test_gpu_fm_keybuild.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2>&1 | grep real
python - $O/bench_n1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g" % d["value"], "ms %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], "wkb %.3f" % d["ms_per_step_with_key_build"],
      "zipf %.4f" % d["zipf"]["ms_per_step"], "fm %.3f" % d["fm"]["ms_per_step"], "sweep", [round(t["ms_per_step"], 3) for t in d["table_sweep"]["tables"]],
      "cpu", round(d["cpu_baseline"]["value"]))
PY
