cd $GRAFT_REPO_ROOT
timeout 1800 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
bash scripts/profile_round.sh r4_v3_call 2>&1 | tail -14
