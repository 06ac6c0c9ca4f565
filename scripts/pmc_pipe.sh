#!/bin/bash
# usage: scripts/pmc_pipe.sh <tag> [bench.py arguments, e.g. --kernel 6]   (run on the GPU box through gpurun)
# Issue / wait / matrix-pipe counters of walk_pipe_kernel over bench.py's resident passes (one launch of 6400 trees per
# pass; --kernel 5: one wave per SIMD, --kernel 6: two), two rocprofv3 --pmc passes (eight SQ counters each), mean per launch
# into gpurun_out/pmc_<tag>.json.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1
X="${@:2}"
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --resident-only $X"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_${T}_a -o a -- $B > $R/gpurun_out/pmc_${T}_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc_${T}_b -o b -- $B > $R/gpurun_out/pmc_${T}_b.log 2>&1
python3 - <<PY
import csv, collections, json
out={}
names=set()
for f in ('$R/gpurun_out/pmc_${T}_a/a_counter_collection.csv','$R/gpurun_out/pmc_${T}_b/b_counter_collection.csv'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); launches=collections.defaultdict(set); per={}; grid={}
    for r in csv.DictReader(open(f)):
        if 'walk_pipe' in r['Kernel_Name']:
            k=r['Kernel_Name'].split('(')[0]
            names.add(k)
            per.setdefault(k,{}).setdefault(r['Dispatch_Id'],collections.defaultdict(float))[r['Counter_Name']]+=float(r['Counter_Value'])
            grid.setdefault(k,{})[r['Dispatch_Id']]=int(r['Grid_Size'])
    # the mean over the FULL-SIZE dispatches only (the largest grid: the resident passes of 6400 trees): the blocking call
    # in front of them runs as two smaller chunks, and round 4's files averaged those in as if they were full launches
    for k in per:
        top=max(grid[k].values())
        for d,c in per[k].items():
            if grid[k][d]==top:
                launches[k].add(d)
                for name,v in c.items(): agg[k][name]+=v
    for k in agg:
        for c,v in agg[k].items(): out.setdefault(k,{})[c]=v/len(launches[k])
        out[k]['launches']=len(launches[k])
for k,o in out.items():
    wc=o.get('SQ_WAVE_CYCLES',0)
    print(k)
    print('  instructions per launch: VALU(incl. MFMA) %.1f M  MFMA %.1f M  LDS %.1f M  SALU %.1f M  SMEM %.1f M  VMEM rd %.2f M wr %.2f M'%tuple(o.get(c,0)/1e6 for c in ('SQ_INSTS_VALU','SQ_INSTS_MFMA','SQ_INSTS_LDS','SQ_INSTS_SALU','SQ_INSTS_SMEM','SQ_INSTS_VMEM_RD','SQ_INSTS_VMEM_WR')))
    if wc:
        print('  wave quad-cycles %.1f M; fractions: active %.3f wait_any %.3f wait_inst %.3f valu %.3f lds %.3f sca %.3f'%((wc/1e6,)+tuple(o.get(c,0)/wc for c in ('SQ_ACTIVE_INST_ANY','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_SCA'))))
        print('  matrix pipe busy cycles %.1f M = %.3f of busy cycles x 4 SIMDs'%(o.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1e6, o.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(o.get('SQ_BUSY_CYCLES',1)*4,1)))
out['_command']='rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --resident-only $X'
json.dump(out, open('$R/gpurun_out/pmc_${T}.json','w'), indent=1)
PY
