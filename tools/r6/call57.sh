#!/bin/bash
# round 6, call 57: memory-side traffic of the hand-written sort's kernels (tools/pmc2.sh: the
# L2's requests to the fabric by size, calibration patterns in the same runs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
bash tools/pmc2.sh gpurun_out/r06g sort_key_pos python $PWD/tools/r6/sort_probe.py 6 --by-hand --first --pmc-calibrate 2>&1 | grep -v calib | cut -c1-200
bash tools/pmc2.sh gpurun_out/r06g sort_key_pos_zipf11 python $PWD/tools/r6/sort_probe.py 6 --by-hand --zipf --pmc-calibrate 2>&1 | grep -v calib | cut -c1-200
rm -f gpurun_out/r06g/*_rd.json gpurun_out/r06g/*_wr.json gpurun_out/r06g/*.err
ls -la gpurun_out/r06g
