cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_many
FUZZ_ONLY=2397 FUZZ_LARGE_TREES=1 timeout 600 python3 scripts/gpu_fuzz.py 2398 6201 1 > gpurun_out/r4_many/case2397.log 2>&1; tail -40 gpurun_out/r4_many/case2397.log
BITO_AMD_PIPE_MIN_BRANCH=0 timeout 600 python3 scripts/gpu_midsize.py 30 32 33 34 36 38 > gpurun_out/r4_many/midsize_many.log 2>&1; tail -12 gpurun_out/r4_many/midsize_many.log
timeout 900 python3 scripts/gpu_fuzz.py 1500 6301 5 > gpurun_out/r4_many/fuzz_pipe_1500_seed6301.log 2>&1; tail -3 gpurun_out/r4_many/fuzz_pipe_1500_seed6301.log
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
