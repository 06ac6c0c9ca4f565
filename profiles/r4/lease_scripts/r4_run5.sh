cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout 600 python3 -m pytest tests/test_time_tree.py tests/test_engine_chunks.py tests/test_cabi_client.py -m gpu -x -q > gpurun_out/r4e/pytest_slots.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4e/pytest_slots.log
tail -15 gpurun_out/r4e/pytest_slots.log
timeout 300 python3 scripts/gpu_slots_timeline.py 8 6400 > gpurun_out/r4e/slots8_threads.log 2>&1; cat gpurun_out/r4e/slots8_threads.log
BITO_AMD_SLOT_THREADS=0 timeout 300 python3 scripts/gpu_slots_timeline.py 8 6400 > gpurun_out/r4e/slots8_one_thread.log 2>&1; cat gpurun_out/r4e/slots8_one_thread.log
timeout 300 python3 scripts/gpu_slots_timeline.py 2 6400 > gpurun_out/r4e/slots2_threads.log 2>&1; cat gpurun_out/r4e/slots2_threads.log
timeout 300 python3 bench.py --engine-devices 0,0 --steps 10 --warmup 3 --no-cpu-baseline --no-resident > gpurun_out/r4e/bench_slots2.json 2> gpurun_out/r4e/bench_slots2.err; tail -2 gpurun_out/r4e/bench_slots2.err; cut -c1-600 gpurun_out/r4e/bench_slots2.json
