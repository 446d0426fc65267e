#!/bin/bash
# Collect several rocprofv3 --pmc passes (one counter group per run, kernel-trace only) of the
# bench and print per-kernel medians.  Usage (on the GPU box, from the repo root):
#   bash tools/pmc_passes.sh <outdir> [extra bench args]
# NOTE: a counter group the hardware cannot schedule makes rocprofv3 abort and then HANG until
# the timeout (it cost 20 GPU-minutes once): keep groups small, keep the hard timeout.
set -u
OUT=$1; shift
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_avr TCC_READ_sum TCC_WRITE_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$OUT/g$i -- \
      python $R/bench.py --steps 6 --warmup 8 --no-cpu-baseline "$@" > $R/$OUT/g$i.json 2> $R/$OUT/g$i.err
done
cd $R
python tools/pmc_summary.py $OUT
