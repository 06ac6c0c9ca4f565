cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
for v in abl_empty abl_skeleton; do
  BITO_AMD_LIB=bito_amd/variants/$v.so timeout 120 python3 scripts/gpu_pipe_ablate.py $v >> gpurun_out/r4c/ablate2.log 2>&1
done
cat gpurun_out/r4c/ablate2.log
bash scripts/pmc_pipe.sh r4_one --kernel 5 > gpurun_out/r4c/pmc_one.log 2>&1; cat gpurun_out/r4c/pmc_one.log
bash scripts/pmc_pipe.sh r4_two --kernel 6 > gpurun_out/r4c/pmc_two.log 2>&1; cat gpurun_out/r4c/pmc_two.log
timeout 300 python3 bench.py --steps 10 --warmup 3 --cpu-seconds 3 > gpurun_out/r4c/bench_auto.json 2> gpurun_out/r4c/bench_auto.err; tail -3 gpurun_out/r4c/bench_auto.err; cat gpurun_out/r4c/bench_auto.json | cut -c1-3000
