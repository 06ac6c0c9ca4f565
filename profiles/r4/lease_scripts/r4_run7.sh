cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gp.py tests/test_nni.py tests/test_tp.py -m gpu -x -q 2>&1 | tail -5
bash scripts/profile_gp_bench.sh r4_gp_ds1_fused ds1 2>&1 | cut -c1-1500
bash scripts/profile_gp_bench.sh r4_gp_seeded_fused seeded 2>&1 | cut -c1-1500
