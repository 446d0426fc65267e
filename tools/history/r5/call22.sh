#!/bin/bash
# round 5, GPU call 22: the FM forward of a replayed minibatch over cells (commit e0b7892, withdrawn in
# d9ca0dd: no faster) against the row-major forward (XF_FM_CELLS=0 in that commit): FM parity
# tests, then the FM bench lines.  Kept for the record; the switch no longer exists.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
O=gpurun_out/r5c22
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fm_keybuild.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_tight.py tests/test_gpu_reforder.py -x -q -k "fm or FM" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -x -q -k "fm or FM" 2>&1 | tail -3
line() {
python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.4f ms" % d["ms_per_step"], d.get("ms_per_step_repeats") and "median %.4f" % d["ms_per_step_repeats"]["median"],
          {k: round(v * 1e3, 1) for k, v in d["kernels_ms"].items() if v}, "frac", round(d["roofline"]["frac"], 3), "wkb", d.get("ms_per_step_with_key_build"))
except Exception as e:
    print(f, "FAILED", e)
PY
}
for C in 1 0; do
XF_FM_CELLS=$C timeout 300 python bench.py --model fm --k 16 --optimizer sgd --no-cpu-baseline --repeats 3 --batches 8 > $O/fm16_c$C.json 2> $O/fm16_c$C.err; line $O/fm16_c$C.json
XF_FM_CELLS=$C timeout 300 python bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --no-cpu-baseline --repeats 3 --batches 8 > $O/fm64_c$C.json 2> $O/fm64_c$C.err; line $O/fm64_c$C.json
done
