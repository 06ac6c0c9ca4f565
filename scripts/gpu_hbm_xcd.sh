#!/bin/bash
# walk_hbm_cat_kernel with and without the XCD-aware tree/tile mapping: config 4 (125 trees) and the mid-size sweep
set -u
out=gpurun_out/hbm_xcd
mkdir -p $out
for m in 0 1; do
  BITO_AMD_HBM_BY_XCD=$m python bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline --no-resident 2>&1 | tail -1 > $out/config4_xcd$m.json
  BITO_AMD_HBM_BY_XCD=$m python scripts/gpu_hbm_sizes.py > $out/sizes_xcd$m.log 2>&1
done
python - <<'PY'
import json
for m in (0, 1):
    d = json.loads(open(f"gpurun_out/hbm_xcd/config4_xcd{m}.json").read())
    print("by_xcd", m, "config4", round(d["value"], 1), "trees/s", round(d["roofline"]["avg_kernel_ms"], 2), "ms per launch")
    print(open(f"gpurun_out/hbm_xcd/sizes_xcd{m}.log").read())
PY
