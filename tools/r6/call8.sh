#!/bin/bash
# round 6, call 8: k_ar_insert without its scratch array; where a large table's defrag spends
# 95 ms (HIP API + kernel trace of the 1e8-key first epoch)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 900 python -m pytest tests/test_gpu_keybuild.py -x -q -m gpu 2>&1 | tail -3
rm -rf /tmp/ft
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -- \
    python "$GRAFT_REPO_ROOT/tools/r6/fresh_probe.py" 10000000 3 30 > /tmp/ft.out 2> /tmp/ft.err)
python tools/r6/timeline.py /tmp/ft 0 60 | grep "k_ar_\|k_kb_scatter\|k_lr_fwd" | cut -c1-100
rm -rf /tmp/ft
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/ft -- \
    python "$GRAFT_REPO_ROOT/tools/r6/fresh_probe.py" 100000000 9 30 > /tmp/ft.out 2> /tmp/ft.err)
tail -1 /tmp/ft.out | cut -c1-600
python tools/r6/api_top.py /tmp/ft 40
timeout 900 python tools/r6/sweep_variants.py 100000000 2>&1 | tail -8 | cut -c1-300
timeout 900 python tools/r6/sweep_variants.py 30000000 2>&1 | tail -8 | cut -c1-300
