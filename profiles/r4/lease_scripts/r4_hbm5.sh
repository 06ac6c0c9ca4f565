cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_hbm
FUZZ_LARGE_TREES=1 timeout 900 python3 scripts/gpu_fuzz.py 500 7101 1 > gpurun_out/r4_hbm/fuzz_fold_500.log 2>&1; tail -3 gpurun_out/r4_hbm/fuzz_fold_500.log
for mode in "BITO_AMD_HBM_R3=1" "BITO_AMD_HBM_FOLD=0" "BITO_AMD_HBM_FOLD=1"; do
echo "== $mode"
env $mode timeout 600 python3 scripts/gpu_hbm_sizes.py 41 64 100 2>&1 | tail -3
env $mode timeout 900 python3 bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline --no-resident 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), j['ms_per_step'], j.get('roofline',{}).get('avg_kernel_ms'), j.get('parity'))"
done
