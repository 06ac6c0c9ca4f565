cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_hbm
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
FUZZ_LARGE_TREES=1 timeout 1500 python3 scripts/gpu_fuzz.py 3000 7201 1 > gpurun_out/r4_hbm/fuzz_hbm_fold_large_3000_seed7201.log 2>&1; tail -2 gpurun_out/r4_hbm/fuzz_hbm_fold_large_3000_seed7201.log
timeout 900 python3 scripts/gpu_fuzz.py 2000 7202 > gpurun_out/r4_hbm/fuzz_any_2000_seed7202.log 2>&1; tail -1 gpurun_out/r4_hbm/fuzz_any_2000_seed7202.log
timeout 600 python3 scripts/gpu_hbm_sizes.py 41 64 70 100 128 > gpurun_out/r4_hbm/hbm_sizes_fold.log 2>&1; cat gpurun_out/r4_hbm/hbm_sizes_fold.log
bash scripts/profile_round.sh r4_v3_config4 --workload config4 --steps 4 --warmup 1 2>&1 | tail -12
