cd $GRAFT_REPO_ROOT
for v in "" w7 w6 w5; do
  if [ -z "$v" ]; then L=bito_amd/libbito_amd.so; else L=bito_amd/variants/libbito_amd_$v.so; fi
  echo "== ${v:-shipped (8 waves)}"
  BITO_AMD_LIB=$L timeout 300 python3 scripts/gpu_config4.py 125 2>&1 | grep "config4"
  BITO_AMD_LIB=$L timeout 300 python3 scripts/gpu_hbm_sizes.py 64 100 2>&1 | grep "n="
done
