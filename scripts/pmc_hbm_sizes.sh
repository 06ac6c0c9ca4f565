#!/bin/bash
# usage: scripts/pmc_hbm_sizes.sh <tag> <taxa ...>   (on the GPU box through gpurun)
# HBM traffic of walk_hbm_cat_kernel on mid-size trees (scripts/gpu_hbm_sizes.py: 1600 trees x 1000 patterns, GTR+weibull4,
# gradient, no rescaling): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (MI355X_MICROARCH.md HBM section: KB
# units, FETCH_SIZE doubled on gfx950), mean per launch of the walk kernel, beside the kernel's time from a plain run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1
shift
for n in "$@"; do
  python3 $R/scripts/gpu_hbm_sizes.py $n > $R/gpurun_out/${T}_${n}_time.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_${n}_fetch -o f -- python3 $R/scripts/gpu_hbm_sizes.py $n > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_${n}_write -o w -- python3 $R/scripts/gpu_hbm_sizes.py $n > /dev/null 2>&1
done
python3 - "$T" "$@" <<PY
import csv, collections, json, re, sys
R='$R'; T=sys.argv[1]
out=[]
for n in sys.argv[2:]:
    vals={}
    for name,f in (('FETCH_SIZE',f'{R}/gpurun_out/{T}_{n}_fetch/f_counter_collection.csv'),('WRITE_SIZE',f'{R}/gpurun_out/{T}_{n}_write/w_counter_collection.csv')):
        v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'walk_hbm_cat' in r['Kernel_Name']]
        vals[name]=sum(v)/len(v)*1024
    line=open(f'{R}/gpurun_out/{T}_{n}_time.log').read()
    ms=float(re.search(r'walk kernel ([\d.]+) ms', line).group(1))
    n_i=int(n); P=1000; C=4; trees=1600
    hbm=2*vals['FETCH_SIZE']+vals['WRITE_SIZE']
    alg=(13*n_i-12)*C*P*4*8*trees
    flops=C*P*((3*n_i-3)*60+(2*n_i-2)*43)*trees
    out.append({'taxa':n_i,'trees_per_launch':trees,'patterns':P,'kernel_ms':ms,'FETCH_SIZE_bytes_raw':vals['FETCH_SIZE'],'WRITE_SIZE_bytes':vals['WRITE_SIZE'],
                'hbm_bytes_per_launch':hbm,'hbm_GBps':hbm/ms/1e6,'hbm_frac_of_8TBps':hbm/ms/1e6/8000,
                'algorithmic_bytes_per_launch':alg,'traffic_over_algorithmic':hbm/alg,
                'algorithmic_TFLOPs':flops/ms/1e9,'frac_of_fp64_peak':flops/ms/1e9/78.6})
    print(out[-1])
json.dump({'command':'scripts/pmc_hbm_sizes.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes, -- python3 scripts/gpu_hbm_sizes.py <taxa>)','gfx950_correction':'FETCH_SIZE doubled, WRITE_SIZE as reported; KB units','rows':out}, open(f'{R}/gpurun_out/{T}_hbm_sizes_pmc.json','w'), indent=1)
PY
