cd $GRAFT_REPO_ROOT
for cfg in "1 1000000" "2 1000000" "4 1000000" "8 1000000" "8 500000" "1 4000000" "8 4000000"; do python tools/sharded_step_timing.py $cfg 2>&1 | tail -1; done
