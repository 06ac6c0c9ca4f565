cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 300 python3 scripts/gpu_pipe.py 2>&1 | head -8
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15
for f in 1 0; do
  echo "== BITO_AMD_PIPE_FOLD=$f"
  BITO_AMD_PIPE_FOLD=$f timeout 200 python3 scripts/gpu_pipe_ablate.py fold$f 2>&1 | tail -2
  BITO_AMD_PIPE_FOLD=$f BITO_AMD_PIPE_MIN_BRANCH=0 timeout 600 python3 scripts/gpu_midsize.py 38 44 50 56 60 64 2>&1 | grep -v "^$" | tail -8
done
