#!/bin/bash
# usage: scripts/pmc_lds.sh <tag>   (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_${T}_a -o a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${T}_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_${T}_b -o b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${T}_b.log 2>&1
python3 - <<PY
import csv, collections
out={}
for f in ('$R/gpurun_out/pmc_${T}_a/a_counter_collection.csv','$R/gpurun_out/pmc_${T}_b/b_counter_collection.csv'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'walk_tree' in r['Kernel_Name'] or 'walk_lds' in r['Kernel_Name'] or 'walk_pipe' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): out[k]=sum(v)/len(v)
trees=int('${TREES:-6400}')  # trees per launch of the bench default
waves=trees*15*4  # trees x 15 tiles of 64 patterns x 4 waves (G = 4 groups of 4 patterns per wave)
wc=out['SQ_WAVE_CYCLES']
steps=2*17.6  # stored (non-cherry) internal nodes per pass, DS1 average
print('per wave-step: instr VALU %.1f MFMA %.1f LDS %.1f SALU %.1f VMEM %.1f'%tuple(out[k]/waves/steps for k in ('SQ_INSTS_VALU','SQ_INSTS_MFMA','SQ_INSTS_LDS','SQ_INSTS_SALU','SQ_INSTS_VMEM_RD')))
print('wave cycles per step (x4): %.0f'%(wc*4/waves/steps))
for k in ('SQ_ACTIVE_INST_ANY','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_SCA','SQ_ACTIVE_INST_VMEM'):
    if k in out: print(k, '%.3f'%(out[k]/wc))
print('MFMA busy cycles per SIMD', out['SQ_VALU_MFMA_BUSY_CYCLES']/1024)
import json
out['_note']='mean per launch of the walk kernel over bench.py launches; %d trees x 15 tiles x 4 waves, about 35 tree steps per wave and tile (cherries folded)'%trees
json.dump(out, open('$R/gpurun_out/pmc_${T}.json','w'), indent=1)
PY
