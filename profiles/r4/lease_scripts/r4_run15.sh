cd $GRAFT_REPO_ROOT
BITO_AMD_PIPE_MIN_BRANCH=0 timeout 900 python3 scripts/gpu_midsize.py 30 38 41 44 48 50 52 56 57 58 60 62 64 2>&1 | grep "kernel=walk_pipe\|kernel=walk_hbm\|vs walk_pipe" > gpurun_out/r4f_midsize_fold.log
cat gpurun_out/r4f_midsize_fold.log
