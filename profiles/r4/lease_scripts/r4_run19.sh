cd $GRAFT_REPO_ROOT
for n in 16 12; do
BITO_AMD_LIB=bito_amd/variants/two16.so BITO_AMD_PIPE_MIN_BRANCH=0 timeout 300 python3 scripts/gpu_pipe_two_small.py $n 2>&1 | tail -3
BITO_AMD_PIPE_MIN_BRANCH=0 timeout 300 python3 scripts/gpu_pipe_two_small.py $n 2>&1 | tail -3
done
