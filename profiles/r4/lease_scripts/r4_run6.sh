cd $GRAFT_REPO_ROOT
bash scripts/profile_gp_bench.sh r4_gp_ds1 ds1
bash scripts/profile_gp_bench.sh r4_gp_seeded seeded
tail -3 gpurun_out/r4_gp_ds1_bench.err
