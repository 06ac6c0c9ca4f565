#!/bin/bash
# usage: scripts/build_hbm_variants.sh name "flags" [name "flags" ...]  -> bito_amd/libbito_amd_<name>.so
# (experiments on walk_hbm_kernel's build-time knobs: HBM_PRE_UNROLL, HBM_WAVES_EXPR)
cd $(dirname $0)/../bito_amd/csrc
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -amdgpu-sched-strategy=max-ilp $flags -c kernels.hip -o /tmp/kernels_$name.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbito_amd_$name.so /tmp/kernels_$name.o gs_kernels.o walk_lds.o walk_tree.o time_tree.o worker.o engine.o beagle_shim.o gp_engine.o || exit 1
  echo built $name
done
