#!/bin/bash
# Builds timing-only variants of the LDS kernel (ABLATE bitmask, see walk_lds.hip) into
# gpurun_variants/ (not product code; results of these builds are wrong by design).
set -e
cd "$(dirname "$0")/../bito_amd/csrc"
mkdir -p ../../gpurun_variants
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -DABLATE=$a -c walk_lds.hip -o /tmp/walk_lds_$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_variants/lib_ablate_$a.so kernels.o /tmp/walk_lds_$a.o engine.o
done
