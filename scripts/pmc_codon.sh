#!/bin/bash
# usage: scripts/pmc_codon.sh <tag>   (run on the GPU box through gpurun)
# issue / wait / matrix-pipe counters of gs_walk_kernel on config 5 (bench.py --workload codon, 4096 trees per launch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1
B="python3 $R/bench.py --workload codon --steps 2 --warmup 1 --no-cpu-baseline --no-resident"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcc_${T}_a -o a -- $B > $R/gpurun_out/pmcc_${T}_a.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmcc_${T}_b -o b -- $B > $R/gpurun_out/pmcc_${T}_b.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcc_${T}_c -o c -- $B > $R/gpurun_out/pmcc_${T}_c.log 2>&1
python3 - <<PY
import csv, collections, json
out={}
for f in ('a','b','c'):
    try:
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f'$R/gpurun_out/pmcc_${T}_{f}/{f}_counter_collection.csv')):
            if 'gs_walk_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items(): out[k]=sum(v)/len(v)
    except Exception as e:
        print('pass',f,'failed:',e)
wc=out.get('SQ_WAVE_CYCLES',1)
for k,v in sorted(out.items()):
    print(k, v, '%.3f of wave cycles'%(v/wc))
json.dump(out, open('$R/gpurun_out/pmcc_${T}.json','w'), indent=1)
PY
