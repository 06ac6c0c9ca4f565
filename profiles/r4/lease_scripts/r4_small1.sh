cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_small
python3 scripts/gpu_small_call_models.py 100
cd /tmp && export TMPDIR=/tmp
for m in GTR+weibull+4 JC69+weibull+4 GTR+constant; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_small/$m -o s -- python3 $GRAFT_REPO_ROOT/scripts/gpu_small_call_models.py 100 $m > /dev/null 2>&1
echo "== $m"; python3 - <<PY
import csv
for r in csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/r4_small/$m/s_kernel_stats.csv')):
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2), 'us')
PY
done
