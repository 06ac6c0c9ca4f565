cd $GRAFT_REPO_ROOT
for v in "" rows; do
  if [ -z "$v" ]; then L=bito_amd/libbito_amd.so; else L=bito_amd/variants/libbito_amd_$v.so; fi
  echo "== ${v:-new layout (32 bytes per lane)}"
  BITO_AMD_LIB=$L timeout 300 python3 scripts/gpu_config4.py 125 2>&1 | grep "config4\|config2"
  BITO_AMD_LIB=$L timeout 300 python3 scripts/gpu_hbm_sizes.py 64 100 2>&1 | grep "n="
done
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
