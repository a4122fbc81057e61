cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r3c/pytest_all.log 2>&1; echo "rc all $?"
tail -4 gpurun_out/r3c/pytest_all.log
