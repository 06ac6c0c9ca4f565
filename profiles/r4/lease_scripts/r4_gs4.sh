# (gs_matrices_kernel's ablations: the variants m1..m4 are builds of commit 41077d6 with -DGS_MAT_EXP=1..4 -- the knobs were
# taken out of gs_kernels.hip again so that the shipped device code is the one the round's last full GPU run checked)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default m1 m2 m3 m4; do
  if [ $v = default ]; then lib=$R/bito_amd/libbito_amd.so; else lib=$R/bito_amd/variants/$v.so; fi
  BENCH_ABLATION=1 BITO_AMD_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_gs/mat_$v -o s -- python3 $R/bench.py --workload codon --steps 3 --warmup 1 --no-cpu-baseline --no-resident --no-parity-check > /dev/null 2>&1
  python3 - <<PY
import csv,os
f='$R/gpurun_out/r4_gs/mat_$v/s_kernel_stats.csv'
if os.path.exists(f):
    for r in csv.DictReader(open(f)):
        if 'gs_matrices' in r['Name'] or 'gs_walk' in r['Name']: print('$v', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
else: print('$v: no stats')
PY
done
