#!/bin/bash
# Alternative libbito_amd.so files that differ only in walk_lds.hip's build-time knobs, into
# bito_amd/variants/<name>.so; run one with BITO_AMD_LIB=bito_amd/variants/<name>.so.
# usage: scripts/build_lds_variants.sh name "-DLDS_WAVES=8 -DLDS_WAVES_PER_EU=2 -DLDS_FORCE_G=2" [name flags ...]
set -e
cd "$(dirname "$0")/../bito_amd/csrc"
make -s
mkdir -p ../variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=iterative-ilp $flags -c walk_lds.hip -o /tmp/wl_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so kernels.o walk_hbm_cat.o gs_kernels.o /tmp/wl_$name.o walk_pipe.o walk_tree.o time_tree.o worker.o engine.o beagle_shim.o gp_engine.o
  echo built $name
done
