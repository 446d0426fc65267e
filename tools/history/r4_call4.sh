#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c4
mkdir -p $O
cd $R
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp/gather_lines.hip -o /tmp/gl 2>/dev/null && /tmp/gl 2>&1 | tee -a $O/summary.txt
echo "--- chunk bits 12 ---" | tee -a $O/summary.txt
XF_LIB=$R/xflow_amd/lib/var_cb12/libxflow_amd.so timeout 300 python tools/cells_knobs.py --knobs 299,304,305,320,336,400,304 --steps 32 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
echo "--- chunk bits 11 ---" | tee -a $O/summary.txt
timeout 300 python tools/cells_knobs.py --knobs 304,320,336,400 --steps 32 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
