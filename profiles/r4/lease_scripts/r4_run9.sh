cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or beagle" 2>&1 | tail -3
timeout 300 python3 -m pytest tests/test_cabi_client.py -m gpu -x -q 2>&1 | tail -3
bash scripts/pmc_hbm_sizes.sh r4 64 100 2>&1 | tail -4
