cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_fuzz
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 1500 python3 scripts/gpu_fuzz.py 6000 6101 > gpurun_out/r4_fuzz/r4_fold_fuzz_any_6000_seed6101.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fold_fuzz_any_6000_seed6101.log
timeout 900 python3 scripts/gpu_fuzz.py 5000 6102 5 > gpurun_out/r4_fuzz/r4_fold_fuzz_pipe_5000_seed6102.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fold_fuzz_pipe_5000_seed6102.log
timeout 900 python3 scripts/gpu_fuzz.py 3000 6103 6 > gpurun_out/r4_fuzz/r4_fold_fuzz_pipe_two_waves_3000_seed6103.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fold_fuzz_pipe_two_waves_3000_seed6103.log
BITO_AMD_CHUNK_FIRST=4 BITO_AMD_CHUNK_GROWTH=2 BITO_AMD_CHUNK_CAP=16 BITO_AMD_HOST_MIN_TREES=4 BITO_AMD_HOST_THREADS=5 BITO_AMD_CHUNK_RESERVE=32 timeout 900 python3 scripts/gpu_fuzz.py 1500 6104 > gpurun_out/r4_fuzz/r4_fold_fuzz_chunked_threads_1500_seed6104.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fold_fuzz_chunked_threads_1500_seed6104.log
timeout 900 python3 scripts/gpu_call_soak.py 400 32 > gpurun_out/r4_fuzz/r4_fold_call_soak_slots_400_seed32.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fold_call_soak_slots_400_seed32.log
