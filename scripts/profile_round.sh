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
# The traversal launches of one blocking call OVERLAP (the second chunk's workgroups move in as the first chunk's leave), so
# the per-launch average of the stats file counts that stretch twice: from the kernel trace of the same run, the spans of
# the walk kernel's launches added up and their UNION, per device, and both per launch -- what bench.py reports as
# avg_launch_span_ms and avg_kernel_ms, reproducible from this file alone.
import glob
trace = glob.glob(f'{R}/gpurun_out/{T}_stats/**/*kernel_trace.csv', recursive=True)
if trace:
    per_dev = collections.defaultdict(list)
    for r in csv.DictReader(open(trace[0])):
        if 'walk_' in r['Kernel_Name'] and 'walk_' in k and r['Kernel_Name'].split('(')[0] == k:
            per_dev[r.get('Agent_Id', '0')].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    spans = {'kernel': k, 'source': 'rocprofv3 --kernel-trace of: ' + out['command'].split(' -- ')[-1].replace('--pmc FETCH_SIZE|WRITE_SIZE (separate passes) ', ''), 'devices': {}}
    for dev, iv in per_dev.items():
        iv.sort()
        total = sum(b - a for a, b in iv)
        union, cur_a, cur_b = 0, None, None
        for a, b in iv:
            if cur_b is None or a > cur_b:
                if cur_b is not None: union += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        if cur_b is not None: union += cur_b - cur_a
        spans['devices'][dev] = {'launches': len(iv), 'sum_of_spans_ms': total / 1e6, 'union_of_spans_ms': union / 1e6,
                                 'avg_launch_span_ms': total / 1e6 / len(iv), 'avg_kernel_ms_union': union / 1e6 / len(iv)}
    json.dump(spans, open(f'{R}/gpurun_out/{T}_kernel_spans.json', 'w'), indent=1)
    print('kernel spans:', json.dumps(spans['devices']))
print(open(f'{R}/gpurun_out/{T}_bench.json').read())
PY
