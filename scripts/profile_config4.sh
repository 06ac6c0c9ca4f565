#!/bin/bash
# usage: scripts/profile_config4.sh <tag> [trees]   (on the GPU box through gpurun)
# BASELINE config 4 shape at one GPU's share (125 trees of 1000 taxa x 10000 patterns, GTR+weibull4,
# rescaling on): timing, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1
N=${2:-125}
B="python3 $R/scripts/gpu_config4.py $N"
$B > $R/gpurun_out/${T}_timing.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -o s -- $B > $R/gpurun_out/${T}_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -o f -- $B > $R/gpurun_out/${T}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -o w -- $B > $R/gpurun_out/${T}_write.log 2>&1
cat $R/gpurun_out/${T}_timing.log
python3 - <<PY
import csv, collections, json
R='$R'; T='$T'
out={'command':'rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 scripts/gpu_config4.py $N','timing':open(f'{R}/gpurun_out/{T}_timing.log').read().splitlines(),'kernels':{}}
agg={}
for name,f in (('FETCH_SIZE',f'{R}/gpurun_out/{T}_fetch/f_counter_collection.csv'),('WRITE_SIZE',f'{R}/gpurun_out/{T}_write/w_counter_collection.csv')):
    a=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        a[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    agg[name]=a
for r in csv.DictReader(open(f'{R}/gpurun_out/{T}_stats/s_kernel_stats.csv')):
    k=r['Name'].split('(')[0]; ms=float(r['AverageNs'])/1e6
    row={'calls':int(r['Calls']),'avg_ms':ms,'percent':float(r['Percentage'])}
    f=agg['FETCH_SIZE'].get(k); w=agg['WRITE_SIZE'].get(k)
    if f and w:
        fb=sum(f)/len(f)*1024; wb=sum(w)/len(w)*1024
        row.update(FETCH_SIZE_bytes_raw=fb, WRITE_SIZE_bytes=wb, hbm_bytes_per_launch=2*fb+wb, hbm_GBps=(2*fb+wb)/ms/1e6)
    out['kernels'][k]=row
    if row['percent']>0.5: print(k[:70], row)
out['gfx950_correction']='FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section), WRITE_SIZE as reported; both in KB; averages over all launches of a kernel'
json.dump(out, open(f'{R}/gpurun_out/{T}_summary.json','w'), indent=1)
PY
