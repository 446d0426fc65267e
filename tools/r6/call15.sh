#!/bin/bash
# round 6, call 15: the new-key count in place of the radix sort (arrival build) and the
# big-cell copy spread over workgroups: tests, then the kernel stats again
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 1500 python -m pytest tests/test_gpu_keybuild.py tests/test_gpu_cells.py tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5
bash tools/profile_round6.sh gpurun_out/r06b nopmc 2>&1 | cut -c1-900
du -sh gpurun_out/r06b; ls gpurun_out/r06b
