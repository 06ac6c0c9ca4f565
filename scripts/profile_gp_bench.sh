#!/bin/bash
# usage: scripts/profile_gp_bench.sh <tag> [ds1|seeded]   (on the GPU box through gpurun)
# Path B's bench line (bench.py --workload gp) and the rocprofv3 kernel stats of the same command.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1
D=${2:-ds1}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -o s -- python3 $R/bench.py --workload gp --gp-dag $D --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${T}_stats.log 2>&1
cp $R/gpurun_out/${T}_stats/s_kernel_stats.csv $R/gpurun_out/${T}_kernel_stats.csv 2>/dev/null
python3 $R/bench.py --workload gp --gp-dag $D --steps 20 --warmup 3 --cpu-seconds 10 > $R/gpurun_out/${T}_bench.json 2> $R/gpurun_out/${T}_bench.err
head -12 $R/gpurun_out/${T}_kernel_stats.csv | cut -c1-200
cat $R/gpurun_out/${T}_bench.json
