#!/bin/bash
# usage: scripts/profile_gp.sh <tag>   (on the GPU box through gpurun)
# rocprofv3 kernel stats of the Path B / NNI-proposal measurement (scripts/gpu_gp.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -o s -- python3 $R/scripts/gpu_gp.py > $R/gpurun_out/${T}_stats.log 2>&1
cp $R/gpurun_out/${T}_stats/s_kernel_stats.csv $R/gpurun_out/${T}_kernel_stats.csv 2>/dev/null
tail -5 $R/gpurun_out/${T}_stats.log
head -8 $R/gpurun_out/${T}_kernel_stats.csv
