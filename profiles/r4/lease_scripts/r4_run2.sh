cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 120 bito_amd/csrc/issue_bench.bin > gpurun_out/r4b/issue_costs.txt 2>&1
# a lone G = 2 wave per SIMD (32-pattern tiles) next to the shipped forms
BITO_AMD_PIPE_GROUPS=2 timeout 120 python3 scripts/gpu_pipe_ablate.py groups2 > gpurun_out/r4b/groups2.log 2>&1
BITO_AMD_PIPE_GROUPS=1 timeout 120 python3 scripts/gpu_pipe_ablate.py groups1 >> gpurun_out/r4b/groups2.log 2>&1
cat gpurun_out/r4b/issue_costs.txt | tail -32; cat gpurun_out/r4b/groups2.log
