cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_hbm
timeout 300 python3 scripts/gpu_config2.py > gpurun_out/r4_hbm/config2.log 2>&1; tail -6 gpurun_out/r4_hbm/config2.log
bash scripts/pmc_hbm_sizes.sh r4_hbm_fold_sizes 64 100 2>&1 | tail -4
cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py --workload config4 --steps 8 --warmup 2 > gpurun_out/r4_v3_config4_bench.json 2> gpurun_out/r4_v3_config4_bench.err; cut -c1-400 gpurun_out/r4_v3_config4_bench.json
timeout 900 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
