#!/bin/bash
# variants of walk_hbm_cat_kernel (scripts/build_hbm_cat_variants.sh) against the default build: config 4 and mid sizes
# usage: scripts/gpu_hbm_variants.sh name [name ...]
set -u
mkdir -p gpurun_out/hbm_variants
for v in default "$@"; do
  if [ $v = default ]; then lib=bito_amd/libbito_amd.so; else lib=bito_amd/variants/libbito_amd_$v.so; fi
  BITO_AMD_LIB=$PWD/$lib python bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline --no-resident 2>&1 | tail -1 > gpurun_out/hbm_variants/config4_$v.json
  echo "== $v: $(python -c "import json;d=json.loads(open('gpurun_out/hbm_variants/config4_$v.json').read());print('config4', round(d['value'],1), 'trees/s, walk', round(d['roofline']['avg_kernel_ms'],2), 'ms')")"
  BITO_AMD_LIB=$PWD/$lib python scripts/gpu_hbm_sizes.py 70 100 2>&1 | tail -2
done
