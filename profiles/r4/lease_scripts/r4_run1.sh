cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 300 python3 scripts/gpu_pipe_two.py 64 > gpurun_out/r4a/pipe_two.log 2>&1; echo "pipe_two rc=$?" >> gpurun_out/r4a/pipe_two.log
for v in abl_nolds abl_nof64 abl_mfma abl_nomfma; do
  BITO_AMD_LIB=bito_amd/variants/$v.so timeout 120 python3 scripts/gpu_pipe_ablate.py $v >> gpurun_out/r4a/ablate.log 2>&1
done
timeout 120 python3 scripts/gpu_pipe_ablate.py shipped >> gpurun_out/r4a/ablate.log 2>&1
timeout 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4a/pytest.log
tail -40 gpurun_out/r4a/pipe_two.log; cat gpurun_out/r4a/ablate.log; tail -5 gpurun_out/r4a/pytest.log
