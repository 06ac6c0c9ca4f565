cd $GRAFT_REPO_ROOT
for fold in 0 1; do
echo "== fold $fold"
BITO_AMD_HBM_FOLD=$fold FUZZ_ONLY=977 FUZZ_LARGE_TREES=1 timeout 600 python3 scripts/gpu_fuzz.py 978 7201 1 2>&1 | grep -v Warning | cut -c1-600 | tail -8
done
