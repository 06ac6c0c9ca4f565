cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r4_v1_call > gpurun_out/r4_v1_call_profile.log 2>&1
tail -12 gpurun_out/r4_v1_call_profile.log | cut -c1-1800
