cd $GRAFT_REPO_ROOT
export BITO_AMD_HBM_VALU=1
for v in default w7 w6 w5 w4 noload nostore nomem w6nomem; do
  if [ $v = default ]; then lib=bito_amd/libbito_amd.so; else lib=bito_amd/variants/libbito_amd_$v.so; fi
  BITO_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline --no-resident --no-parity-check 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', round(j['value'],1), 'trees/s; walk', round(j['roofline']['avg_kernel_ms'],2), 'ms')"
done
