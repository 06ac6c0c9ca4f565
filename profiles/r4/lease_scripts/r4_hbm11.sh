cd $GRAFT_REPO_ROOT
run() { timeout 300 python3 bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline --no-resident --no-parity-check 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'trees/s; walk', round(j['roofline']['avg_kernel_ms'],2), 'ms')"; }
echo default; run
echo by_xcd=0; BITO_AMD_HBM_BY_XCD=0 run
for v in ld0 st0 aux0; do echo $v; BITO_AMD_LIB=$PWD/bito_amd/variants/libbito_amd_$v.so run; done
