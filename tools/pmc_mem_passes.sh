#!/bin/bash
# rocprofv3 --pmc passes over the vector-memory path (TA / TCP / address translation) of the
# bench's kernels, one counter group per run, kernel-trace only.  Usage (GPU box, repo root):
#   bash tools/pmc_mem_passes.sh <outdir> [extra bench args]
# NOTE: the first version of this script asked for 6-7 TA/TCP counters per pass; rocprofv3 answered
# "exceeds the capabilities of the hardware", aborted and hung until the timeout (4 x 300 s).
# Small groups and a hard 90 s limit since then.
set -u
OUT=$1; shift
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$OUT/g$i -- \
      python $R/bench.py --steps 6 --warmup 8 --no-cpu-baseline --key-build-steps 0 "$@" > $R/$OUT/g$i.json 2> $R/$OUT/g$i.err
done
cd $R
python tools/pmc_summary.py $OUT
