cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py --workload codon --steps 4 --warmup 1 --cpu-seconds 10 > gpurun_out/r4_codon_bench.json 2> gpurun_out/r4_codon_bench.err; tail -2 gpurun_out/r4_codon_bench.err; cut -c1-1800 gpurun_out/r4_codon_bench.json
timeout 900 python3 bench.py --workload config4 --steps 4 --warmup 1 --cpu-seconds 15 > gpurun_out/r4_config4_bench.json 2> gpurun_out/r4_config4_bench.err; tail -2 gpurun_out/r4_config4_bench.err; cut -c1-1800 gpurun_out/r4_config4_bench.json
timeout 300 python3 scripts/gpu_call_latency.py 1 100 400 1600 2>&1 | tail -4
