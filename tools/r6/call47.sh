#!/bin/bash
# round 6, call 47: the worker side of the exchange for LR without the minibatch's other views
# (xf::batch_compile_lr_dev: sort with the row as payload, heads, cells by counting): the sharded
# tests, the sequential-schedule cycle, its kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 900 python -m pytest tests/test_gpu_keybuild.py -m gpu -x -q -k "sort_key_pos" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_world8_fullsize.py -m gpu -x -q 2>&1 | tail -8
bash tools/r6/call46.sh 2>&1 | tail -32
