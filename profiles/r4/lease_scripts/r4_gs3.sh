cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_general.py tests/test_codon_fixtures.py tests/test_model_cache.py -x -q 2>&1 | tail -3
FUZZ_CODON=1 timeout 900 python3 scripts/gpu_fuzz.py 600 8401 2>&1 | tail -1 | cut -c1-60
timeout 600 python3 bench.py --workload codon --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), 'trees/s blocking;', round(j['ms_per_step'],2), 'ms; resident', round(j['resident']['trees_per_s']), 'walk', j['roofline']['avg_kernel_ms'], j['parity']['max_dll'], j['parity']['max_dgrad'])"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_gs/swz_stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --workload codon --steps 4 --warmup 1 --no-cpu-baseline --no-resident --no-parity-check > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/r4_gs/swz_stats/s_kernel_stats.csv')):
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
PY
