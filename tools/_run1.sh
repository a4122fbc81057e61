cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "block_tables or tables_are_built or stages" 2>&1 | tail -3
python tools/ab_r3.py --variants "block_ordered=0" "block_ordered=1" "block_ordered=1 block_e=16" "block_ordered=0 block_e=16" 2>&1 | grep variant | cut -c1-175
python tools/ab_r3.py --lattice-rows --variants "block_ordered=0" "block_ordered=1" 2>&1 | grep variant | cut -c1-175
python tools/ab_r3.py --n 4000000 --variants "block_ordered=0" "block_ordered=1" 2>&1 | grep variant | cut -c1-175
python tools/ab_r3.py --ell 0.6931 --variants "block_ordered=0" "block_ordered=1" 2>&1 | grep variant | cut -c1-175
