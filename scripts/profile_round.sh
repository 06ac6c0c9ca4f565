#!/bin/bash
# usage: scripts/profile_round.sh <tag> [bench.py arguments, e.g. --workload config4 --steps 4 --warmup 1]   (run on the GPU box through gpurun)
# Three separate rocprofv3 runs of the default bench command: kernel-trace stats, then one
# PMC pass each for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md HBM section: separate
# passes, FETCH_SIZE doubled on gfx950, units KB).  Summaries land in gpurun_out/<tag>_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1
# (round 3: bench.py's timed region is the blocking call, five chunked launches of the walk kernel per step; the second
# region -- passes over a resident batch -- is left out so that the per-kernel averages are those of `value`'s region)
X="${@:2}"
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-resident $X"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -o s -- $B > $R/gpurun_out/${T}_stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -o f -- $B > $R/gpurun_out/${T}_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -o w -- $B > $R/gpurun_out/${T}_write.log 2>&1
python3 $R/bench.py --steps 50 --warmup 5 $X > $R/gpurun_out/${T}_bench.json 2> $R/gpurun_out/${T}_bench.err
python3 - <<PY
import csv, collections, json
R='$R'; T='$T'
out={'command':'rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-resident $X','counters':{}}
for name,f in (('FETCH_SIZE',f'{R}/gpurun_out/{T}_fetch/f_counter_collection.csv'),('WRITE_SIZE',f'{R}/gpurun_out/{T}_write/w_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    out['counters'][name]={k:{'launches':len(v),'mean_KB':sum(v)/len(v)} for k,v in agg.items()}
def walk(name):
    c=out['counters'][name]
    k=[k for k in c if 'walk_' in k]
    return k[0], c[k[0]]['mean_KB']*1024
k,fetch=walk('FETCH_SIZE'); _,write=walk('WRITE_SIZE')
out.update(kernel=k, FETCH_SIZE_bytes_raw=fetch, WRITE_SIZE_bytes=write, hbm_bytes_per_launch=2*fetch+write,
           gfx950_correction='FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section), WRITE_SIZE as reported; both in KB')
json.dump(out, open(f'{R}/gpurun_out/{T}_pmc.json','w'), indent=1)
print(k, 'fetch(raw) %.1f MB  write %.1f MB  hbm/launch %.1f MB'%(fetch/1e6, write/1e6, (2*fetch+write)/1e6))
for r in csv.DictReader(open(f'{R}/gpurun_out/{T}_stats/s_kernel_stats.csv')):
    print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
print(open(f'{R}/gpurun_out/{T}_bench.json').read())
PY
