#!/bin/bash
# Variants of libbito_amd.so that differ in walk_pipe.hip only (generator knobs in the environment, compiler
# flags as the second argument), into bito_amd/variants/<name>.so; run one with BITO_AMD_LIB=...
# usage: scripts/build_pipe_variants.sh name "ENV=1 ..." "-DFLAG ..." [name env flags ...]
set -e
cd "$(dirname "$0")/../bito_amd/csrc"
mkdir -p ../variants /tmp/pipe_variants
while [ $# -ge 3 ]; do
  name=$1; envs=$2; flags=$3; shift 3
  d=/tmp/pipe_variants/$name; mkdir -p $d
  cp walk_pipe.hip kernels.hpp model.hpp $d/
  (cd ../.. && env $envs python3 scripts/gen_walk_pipe.py >/dev/null && cp bito_amd/csrc/walk_pipe_gen.inc $d/)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form $flags -c $d/walk_pipe.hip -o $d/walk_pipe.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so kernels.o gs_kernels.o walk_lds.o $d/walk_pipe.o walk_tree.o time_tree.o worker.o engine.o beagle_shim.o gp_engine.o
  echo built $name
done
(cd ../.. && python3 scripts/gen_walk_pipe.py >/dev/null)  # restore the shipped .inc
