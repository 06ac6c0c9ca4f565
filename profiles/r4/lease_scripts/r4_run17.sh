cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r4_v2_call > gpurun_out/r4_v2_call_profile.log 2>&1
tail -6 gpurun_out/r4_v2_call_profile.log | cut -c1-900
bash scripts/pmc_pipe.sh r4_v2_one --kernel 5 2>&1 | tail -5
