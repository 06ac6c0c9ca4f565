cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_general.py tests/test_codon_fixtures.py tests/test_gpu_parity.py -x -q -k "general or codon or gs or model_index" 2>&1 | tail -3
FUZZ_CODON=1 timeout 900 python3 scripts/gpu_fuzz.py 600 8201 2>&1 | tail -1 | cut -c1-60
for mc in 1 0; do
BITO_AMD_MODEL_CACHE=$mc timeout 600 python3 bench.py --workload codon --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cache $mc:', round(j['value']), 'trees/s blocking;', round(j['ms_per_step'],2), 'ms; resident', round(j['resident']['trees_per_s']), j['parity']['max_dll'], j['parity']['max_dgrad'])"
done
