cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_hbm
for v in default d7 d6; do
  if [ $v = default ]; then lib=bito_amd/libbito_amd.so; else lib=bito_amd/variants/libbito_amd_$v.so; fi
  echo "== $v"
  BITO_AMD_LIB=$PWD/$lib FUZZ_LARGE_TREES=1 timeout 900 python3 scripts/gpu_fuzz.py 300 7301 1 2>&1 | tail -1 | cut -c1-60
  BITO_AMD_LIB=$PWD/$lib timeout 600 python3 scripts/gpu_hbm_sizes.py 64 100 2>&1 | tail -2
  BITO_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline --no-resident --no-parity-check 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'trees/s; walk', round(j['roofline']['avg_kernel_ms'],2), 'ms')"
done
