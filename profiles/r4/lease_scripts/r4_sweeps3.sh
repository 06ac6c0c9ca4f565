cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_fuzz
timeout 1500 python3 scripts/gpu_fuzz.py 8000 8101 > gpurun_out/r4_fuzz/r4_end_fuzz_any_8000_seed8101.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_end_fuzz_any_8000_seed8101.log
timeout 900 python3 scripts/gpu_fuzz.py 4000 8102 5 > gpurun_out/r4_fuzz/r4_end_fuzz_pipe_4000_seed8102.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_end_fuzz_pipe_4000_seed8102.log
FUZZ_LARGE_TREES=1 timeout 900 python3 scripts/gpu_fuzz.py 3000 8103 1 > gpurun_out/r4_fuzz/r4_end_fuzz_hbm_large_3000_seed8103.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_end_fuzz_hbm_large_3000_seed8103.log
timeout 900 python3 scripts/gpu_call_soak.py 2000 41 > gpurun_out/r4_fuzz/r4_end_call_soak_slots_2000_seed41.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_end_call_soak_slots_2000_seed41.log
