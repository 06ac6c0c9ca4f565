#!/bin/bash
# Variants of libbito_amd.so that differ in walk_pipe.hip only (generator knobs in the environment, compiler
# flags as the second argument), into bito_amd/variants/<name>.so; run one with BITO_AMD_LIB=...
# usage: scripts/build_pipe_variants.sh name "ENV=1 ..." "-DFLAG ..." [name env flags ...]
# (generator knobs for timing-only builds, results are wrong by design: PIPE_NO_LDS, PIPE_DROP_F64, PIPE_DROP_TIPS,
# PIPE_DROP_MFMA, PIPE_NO_MFMA, PIPE_NO_SINK -- scripts/gen_walk_pipe.py)
set -e
cd "$(dirname "$0")/../bito_amd/csrc"
make -s walk_pipe_two.csv
mkdir -p ../variants /tmp/pipe_variants
while [ $# -ge 3 ]; do
  name=$1; envs=$2; flags=$3; shift 3
  d=/tmp/pipe_variants/$name; mkdir -p $d
  cp walk_pipe.hip kernels.hpp model.hpp walk_pipe_two.csv $d/
  (cd ../.. && env $envs python3 scripts/gen_walk_pipe.py >/dev/null && cp bito_amd/csrc/walk_pipe_gen.inc $d/)
  # (the two-wave kernels' AGPR allocation follows the variant's own image-register count)
  sed -i "s/amdgpu-agpr-alloc=[0-9]*/amdgpu-agpr-alloc=$(sed -n 's/^#define WALK_PIPE_T_IMAGE_REGS //p' $d/walk_pipe_gen.inc)/" $d/walk_pipe_two.csv
  (cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -mllvm -forceattrs-csv-path=walk_pipe_two.csv $flags -c walk_pipe.hip -o walk_pipe.o 2>&1 | grep -v "^Function in CSV file" || true)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so kernels.o walk_hbm_cat.o gs_kernels.o walk_lds.o $d/walk_pipe.o walk_tree.o time_tree.o worker.o engine.o beagle_shim.o gp_engine.o
  echo built $name
done
(cd ../.. && python3 scripts/gen_walk_pipe.py >/dev/null)  # restore the shipped .inc
