cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_mx
export BITO_AMD_HBM_VALU=0
timeout 900 python3 scripts/gpu_fuzz.py 400 7001 1 > gpurun_out/r4_mx/fuzz_mx_400.log 2>&1; tail -4 gpurun_out/r4_mx/fuzz_mx_400.log
for v in 1 0; do
export BITO_AMD_HBM_VALU=$v
echo "== BITO_AMD_HBM_VALU=$v"
timeout 600 python3 scripts/gpu_hbm_sizes.py 41 64 100 2>&1 | tail -3
timeout 900 python3 bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j.get('roofline',{}).get('avg_kernel_ms'), j.get('parity'))"
done
