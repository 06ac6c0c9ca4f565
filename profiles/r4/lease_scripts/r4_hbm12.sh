cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_hbm
FUZZ_LARGE_TREES=1 timeout 900 python3 scripts/gpu_fuzz.py 600 7401 1 2>&1 | tail -1 | cut -c1-80
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "four_groups or hbm or config4" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_hbm/order_stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline --no-resident --no-parity-check > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/r4_hbm/order_stats/s_kernel_stats.csv')):
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
cd $GRAFT_REPO_ROOT
timeout 300 python3 bench.py --workload config4 --steps 6 --warmup 2 --no-cpu-baseline --no-resident 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'trees/s; step', round(j['ms_per_step'],2), 'walk', round(j['roofline']['avg_kernel_ms'],2), 'ms', j['parity']['max_dll_scaled'], j['parity']['max_dgrad_relative'])"
