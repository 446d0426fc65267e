#!/bin/bash
# round 6, call 40: k_cells_sort_pos — cells of up to 64 entries ranked in registers (the N = 8
# owner shape's 2e5 cells of ~50 entries): tests, the shape's compile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_cells.py tests/test_gpu_sharded.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
N8="--rows 400000 --nnz-per-row 25 --keys-per-gpu 12500000 --force-sharded --general-path --schedule owner --no-cpu-baseline"
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_n -- python $GRAFT_REPO_ROOT/bench.py $N8 --signal-keys 0 --steps 4 --warmup 2 --repeats 0 --batches 2 --no-owner-leg --key-build-steps 16 > /tmp/_n.out 2>&1)
F=$(find /tmp/_n -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_cells_sort_pos", "k_kb_resolve", "k_lr_fwd_cells")):
        print(r["Name"][:52], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
tail -1 /tmp/_n.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n8 compile+step', d.get('ms_per_step'), d.get('ms_per_step_with_key_build'), d.get('with_key_build'))"
Q="--no-cpu-baseline --no-fresh-table --no-n8-shape --no-end-to-end --sustained-seconds 0 --no-fm-leg --no-zipf-leg --no-table-sweep"
python bench.py $Q --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n1', d['ms_per_step'], d.get('ms_per_step_with_key_build'), {k:round(v*1e3,1) for k,v in d['kernels_ms'].items() if v})"
