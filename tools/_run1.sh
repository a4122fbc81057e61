cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "structure or stages or golden or sharded" 2>&1 | tail -2
python tools/ab_build_r3.py --variants "" 2>&1 | grep variant
python tools/ab_build_r3.py --ell 0.25 --variants "" 2>&1 | grep variant
python tools/ab_loop_r3.py --variants "" 2>&1 | grep variant
