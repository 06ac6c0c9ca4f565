cd $GRAFT_REPO_ROOT
for sp in 1 2 4 8 16; do
echo "== image split $sp"
BITO_AMD_PIPE_IMAGE_SPLIT=$sp python3 scripts/gpu_small_call_models.py 100 GTR+weibull+4
BITO_AMD_PIPE_IMAGE_SPLIT=$sp python3 scripts/gpu_small_call_models.py 1 GTR+weibull+4
done
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_engine_chunks.py tests/test_model_cache.py -x -q 2>&1 | tail -2
timeout 600 python3 scripts/gpu_fuzz.py 1500 8301 5 2>&1 | tail -1 | cut -c1-70
