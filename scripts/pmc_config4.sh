#!/bin/bash
# usage: scripts/pmc_config4.sh <tag> [trees]   (on the GPU box through gpurun)
# issue / wait counters of walk_hbm_kernel on the config-4 shape (1000 taxa x 10000 patterns, rescaling on)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1
N=${2:-25}
B="python3 $R/scripts/gpu_config4.py $N"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc4_${T}_a -o a -- $B > $R/gpurun_out/pmc4_${T}_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $R/gpurun_out/pmc4_${T}_b -o b -- $B > $R/gpurun_out/pmc4_${T}_b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $R/gpurun_out/pmc4_${T}_c -o c -- $B > $R/gpurun_out/pmc4_${T}_c.log 2>&1
python3 - <<PY
import csv, collections, json
out={}
for f in ('a','b','c'):
    try:
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f'$R/gpurun_out/pmc4_${T}_{f}/{f}_counter_collection.csv')):
            if 'walk_hbm' in r['Kernel_Name']:
                agg['grad' if 'Lb1ELb1' in r['Kernel_Name'] or 'true, true' in r['Kernel_Name'] else 'll'][r['Counter_Name']].append(float(r['Counter_Value']))
        for kind,d in agg.items():
            for k,v in d.items(): out.setdefault(kind,{})[k]=sum(v)/len(v)
    except Exception as e:
        print('pass',f,'failed:',e)
for kind,d in out.items():
    wc=d.get('SQ_WAVE_CYCLES')
    print(kind, {k:(round(v/wc,3) if wc and k.startswith('SQ_') and k not in ('SQ_WAVE_CYCLES','SQ_WAVES') and not k.startswith('SQ_INSTS') else v) for k,v in d.items()})
json.dump(out, open('$R/gpurun_out/pmc4_${T}.json','w'), indent=1)
PY
