#!/bin/bash
# Kernels that spill registers or use scratch memory (no GPU needed: hipcc's resource-usage
# remarks for gfx950).  Round 6: k_sp_sort's 64-register variant spilled 13 registers — 95 MB of
# scratch writes per 1e7-key sort, which the L2's write-request counters showed as 216 MB
# written for 120 MB of output (profiles/r06/pmc_traffic_sort_key_pos.json).
#   bash tools/spill_check.sh [file.hip ...]        (default: every .hip of xflow_amd/csrc)
R=$(cd "$(dirname "$0")/.." && pwd)
FILES=${@:-$R/xflow_amd/csrc/*.hip}
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
      -I$R/include -I$R/xflow_amd/csrc -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c "
import re, sys
cur, info = None, {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1); info[cur] = {}
    for k in ('VGPRs:', 'ScratchSize [bytes/lane]:', 'VGPRs Spill:', 'Occupancy [waves/SIMD]:'):
        if k in line and cur:
            info[cur][k.rstrip(':')] = int(re.search(re.escape(k) + r'\s*(\d+)', line).group(1))
for k, v in info.items():
    if v.get('ScratchSize [bytes/lane]', 0) or v.get('VGPRs Spill', 0):
        print('$(basename $f)', k[:100], v)
"
done
