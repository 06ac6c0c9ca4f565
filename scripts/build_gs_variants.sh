#!/bin/bash
# Builds alternative libbito_amd.so files that differ only in gs_kernels.hip's build-time knobs, into
# bito_amd/variants/<name>.so; run one with BITO_AMD_LIB=bito_amd/variants/<name>.so.
# usage: scripts/build_gs_variants.sh name "-DGS_WAVES=1 -DGS_SCHED_BARRIER=0" [name flags ...]
set -e
cd "$(dirname "$0")/../bito_amd/csrc"
make -s
mkdir -p ../variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form $flags -c gs_kernels.hip -o /tmp/gs_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so kernels.o walk_hbm_cat.o /tmp/gs_$name.o walk_lds.o walk_pipe.o walk_tree.o time_tree.o worker.o engine.o beagle_shim.o gp_engine.o
  echo built $name
done
