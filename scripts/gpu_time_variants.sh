#!/bin/bash
for f in bito_amd/libbito_amd.so gpurun_variants/*.so; do
  echo "== $f"
  BITO_AMD_LIB=$PWD/$f python scripts/gpu_time.py 2>&1 | grep walk_lds
done
