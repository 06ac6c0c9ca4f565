cd $GRAFT_REPO_ROOT
export BITO_AMD_HBM_VALU=1
for us in 0 500 1500 2800 4000; do
  BITO_AMD_HBM_STAGGER_US=$us timeout 300 python3 bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline --no-resident --no-parity-check 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('stagger $us us:', round(j['value'],1), 'trees/s; walk', round(j['roofline']['avg_kernel_ms'],2), 'ms')"
done
