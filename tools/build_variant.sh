#!/bin/bash
# Experiment: a second libxflow_amd.so built with extra compiler flags into
# xflow_amd/lib/var_<name>/ (loaded with XF_LIB=<path>; the source-hash check is off then).
#   tools/build_variant.sh <name> <flags...>
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/xflow_amd/lib/var_$name
mkdir -p $out
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$R/include -I$R/xflow_amd/csrc"
objs=""
for s in xf_table.hip xf_model.hip xf_calib.hip xf_batch_dev.hip xf_cells.hip xf_keybuild.hip xf_io.cc xf_batch.cc xf_metrics.cc xf_worker.cc xf_group.cc xf_modelfile.cc xf_sharded.hip xf_ingest.hip; do
  x=""; case $s in *.hip) x="-x hip";; esac
  /opt/rocm/bin/hipcc $FL "$@" $x -c $R/xflow_amd/csrc/$s -o $out/$s.o &
  objs="$objs $out/$s.o"
done
/opt/rocm/bin/hipcc $FL -c $R/xflow_amd/lib/xf_source_hash.cc -o $out/hash.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libxflow_amd.so $objs $out/hash.o
rm -f $out/*.o
echo built $out/libxflow_amd.so
