cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_general.py tests/test_codon_fixtures.py -x -q 2>&1 | tail -3
for ov in 0 2 4 8; do
echo "== BITO_AMD_GS_OVERLAP=$ov"
BITO_AMD_GS_OVERLAP=$ov timeout 600 python3 bench.py --workload codon --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), 'trees/s blocking;', round(j['ms_per_step'],2), 'ms; resident', round(j['resident']['trees_per_s']), j['parity']['max_dll'], j['parity']['max_dgrad'])"
done
