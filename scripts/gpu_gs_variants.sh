#!/bin/bash
# times build variants of gs_kernels.hip (scripts/build_gs_variants.sh) on config 5: trees/s and ms of a blocking call, ms of the
# walk kernel per launch, ms of a resident pass (the difference to the walk is the set-up: model, eigensystem, matrices)
for v in "$@"; do printf "== %s: " $v; BENCH_ABLATION=1 BITO_AMD_LIB=bito_amd/variants/$v.so python bench.py --workload codon --no-cpu-baseline --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['resident']['ms_per_step'])"; done
