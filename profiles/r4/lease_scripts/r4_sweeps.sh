cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_fuzz
timeout 1500 python3 scripts/gpu_fuzz.py 6000 5101 > gpurun_out/r4_fuzz/r4_fuzz_any_6000_seed5101.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fuzz_any_6000_seed5101.log
timeout 900 python3 scripts/gpu_fuzz.py 4000 5102 6 > gpurun_out/r4_fuzz/r4_fuzz_pipe_two_waves_4000_seed5102.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fuzz_pipe_two_waves_4000_seed5102.log
timeout 900 python3 scripts/gpu_fuzz.py 2000 5103 5 > gpurun_out/r4_fuzz/r4_fuzz_pipe_2000_seed5103.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fuzz_pipe_2000_seed5103.log
BITO_AMD_PIPE_TWO=1 timeout 900 python3 scripts/gpu_fuzz.py 3000 5104 > gpurun_out/r4_fuzz/r4_fuzz_any_auto_two_waves_3000_seed5104.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fuzz_any_auto_two_waves_3000_seed5104.log
timeout 900 python3 scripts/gpu_call_soak.py 400 31 > gpurun_out/r4_fuzz/r4_call_soak_slots_400_seed31.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_call_soak_slots_400_seed31.log
timeout 600 python3 scripts/gpu_gp_fuzz.py 600 5204 > gpurun_out/r4_fuzz/r4_fuzz_gp_600_seed5204.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fuzz_gp_600_seed5204.log
