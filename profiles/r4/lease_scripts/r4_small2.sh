cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_model_cache.py tests/test_engine_chunks.py tests/test_cabi_client.py -x -q 2>&1 | tail -5
python3 scripts/gpu_small_call_models.py 100
BITO_AMD_MODEL_CACHE=0 python3 scripts/gpu_small_call_models.py 100 GTR+weibull+4
python3 scripts/gpu_small_call_models.py 1 GTR+weibull+4
timeout 600 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), j['blocking_call_ms'], j['parity']['max_dll'], j['parity']['max_dgrad'])"
