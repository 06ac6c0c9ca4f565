cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_many
BITO_AMD_PIPE_GROUPS=2 BITO_AMD_PIPE_MIN_BRANCH=0 timeout 600 python3 scripts/gpu_midsize.py 33 34 36 38 > gpurun_out/r4_many/midsize_g2.log 2>&1; grep "walk_pipe_kernel:" gpurun_out/r4_many/midsize_g2.log
BITO_AMD_PIPE_MIN_BRANCH=0 timeout 600 python3 scripts/gpu_midsize.py 33 34 35 36 37 38 > gpurun_out/r4_many/midsize_many2.log 2>&1; grep "walk_pipe_kernel:" gpurun_out/r4_many/midsize_many2.log
