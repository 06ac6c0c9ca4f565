cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4d/pytest.log
tail -30 gpurun_out/r4d/pytest.log
