cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "four_groups or hbm or config4" 2>&1 | tail -30
