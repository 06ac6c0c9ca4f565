#!/bin/bash
# usage: scripts/build_hbm_cat_variants.sh name "flags" [name "flags" ...]  -> bito_amd/variants/libbito_amd_<name>.so
# (experiments on walk_hbm_cat_kernel's build-time knobs, e.g. -DHBM_CAT_WAVES=6; run with BITO_AMD_LIB=<path>)
cd $(dirname $0)/../bito_amd/csrc
mkdir -p ../variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $flags -c walk_hbm_cat.hip -o /tmp/walk_hbm_cat_$name.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libbito_amd_$name.so kernels.o /tmp/walk_hbm_cat_$name.o gs_kernels.o walk_lds.o walk_pipe.o walk_tree.o time_tree.o worker.o engine.o beagle_shim.o gp_engine.o || exit 1
  echo built $name
done
