#!/bin/bash
# Experiment helper: build libxflow_amd variants with different cells constants into
# xflow_amd/lib/variants/<name>.so (select one with XF_LIB=<path>).
#   bash tools/build_variants.sh name1 "-DXF_CHUNK_BITS=11" name2 "-DXF_WIN_MAX=8704" ...
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/xflow_amd/csrc
mkdir -p $R/xflow_amd/lib/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$R/include -I$C"
while [ $# -gt 1 ]; do
  name=$1; defs=$2; shift 2
  d=/tmp/xfv_$name; mkdir -p $d
  for f in xf_table.hip xf_model.hip xf_calib.hip xf_batch_dev.hip xf_cells.hip xf_sharded.hip; do
    /opt/rocm/bin/hipcc $FLAGS $defs -x hip -c $C/$f -o $d/$f.o &
  done
  for f in xf_io.cc xf_batch.cc xf_metrics.cc xf_worker.cc xf_group.cc xf_modelfile.cc; do
    /opt/rocm/bin/hipcc $FLAGS $defs -c $C/$f -o $d/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/xflow_amd/lib/variants/$name.so $d/*.o
  echo built $name
done
