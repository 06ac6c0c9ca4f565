cd $GRAFT_REPO_ROOT
for cf in 1024 512 256 128; do
echo "== BITO_AMD_CHUNK_FIRST=$cf"
for T in 400 800 1600 3200 6400; do
BITO_AMD_CHUNK_FIRST=$cf python3 scripts/gpu_small_call_models.py $T GTR+weibull+4
done
done
