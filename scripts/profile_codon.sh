#!/bin/bash
# usage: scripts/profile_codon.sh <tag> [trees]   (on the GPU box through gpurun)
# rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE passes for the config-5 (codon) workload.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1
N=${2:-4096}
B="python3 $R/bench.py --workload codon --trees $N --steps 10 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -o s -- $B > $R/gpurun_out/${T}_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -o f -- $B > $R/gpurun_out/${T}_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -o w -- $B > $R/gpurun_out/${T}_write.log 2>&1
python3 - <<PY
import csv, collections, json
R='$R'; T='$T'
out={'command':'rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --workload codon --trees $N --steps 10 --warmup 2 --no-cpu-baseline','kernels':{}}
stats={}
for r in csv.DictReader(open(f'{R}/gpurun_out/{T}_stats/s_kernel_stats.csv')):
    stats[r['Name'].split('(')[0]]=(int(r['Calls']), float(r['AverageNs'])/1e6, float(r['Percentage']))
agg={}
for name,f in (('FETCH_SIZE',f'{R}/gpurun_out/{T}_fetch/f_counter_collection.csv'),('WRITE_SIZE',f'{R}/gpurun_out/{T}_write/w_counter_collection.csv')):
    a=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        a[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    agg[name]=a
for k,(calls,ms,pct) in stats.items():
    f=agg['FETCH_SIZE'].get(k); w=agg['WRITE_SIZE'].get(k)
    row={'calls':calls,'avg_ms':ms,'percent':pct}
    if f and w:
        fb=sum(f)/len(f)*1024; wb=sum(w)/len(w)*1024
        row.update(FETCH_SIZE_bytes_raw=fb, WRITE_SIZE_bytes=wb, hbm_bytes_per_launch=2*fb+wb, hbm_GBps=(2*fb+wb)/ms/1e6)
    out['kernels'][k]=row
    print(k[:60], row)
out['gfx950_correction']='FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section), WRITE_SIZE as reported; both in KB; averages over all launches of a kernel '
json.dump(out, open(f'{R}/gpurun_out/{T}_summary.json','w'), indent=1)
import subprocess
subprocess.run(f"python3 {R}/bench.py --workload codon --trees $N --steps 20 --warmup 3 > {R}/gpurun_out/{T}_bench.json 2> {R}/gpurun_out/{T}_bench.err", shell=True)
print(open(f'{R}/gpurun_out/{T}_bench.json').read())
PY
