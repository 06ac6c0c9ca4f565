cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_fuzz
FUZZ_LARGE_TREES=1 timeout 1500 python3 scripts/gpu_fuzz.py 3000 6201 1 > gpurun_out/r4_fuzz/r4_fold_fuzz_hbm_large_3000_seed6201.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fold_fuzz_hbm_large_3000_seed6201.log
FUZZ_CODON=1 timeout 1500 python3 scripts/gpu_fuzz.py 1500 6202 > gpurun_out/r4_fuzz/r4_fold_fuzz_codon_1500_seed6202.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fold_fuzz_codon_1500_seed6202.log
timeout 900 python3 scripts/gpu_gp_fuzz.py 600 6203 > gpurun_out/r4_fuzz/r4_fold_fuzz_gp_600_seed6203.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fold_fuzz_gp_600_seed6203.log
timeout 900 python3 scripts/gpu_call_soak.py 2000 33 > gpurun_out/r4_fuzz/r4_fold_call_soak_slots_2000_seed33.log 2>&1; tail -1 gpurun_out/r4_fuzz/r4_fold_call_soak_slots_2000_seed33.log
