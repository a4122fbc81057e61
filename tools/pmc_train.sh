#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE / L2 hit-miss counters of the kernels of one training step
# (tools/train_step_r4.py --pre $PRE), one counter group per pass.  FILTER selects the kernels printed.
#   PRE=0 FILTER="slice_contract|splat_wide|blur_axis_multi|backward_pack" TAG=r05_train_pmc tools/pmc_train.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); O=$REPO/gpurun_out/${TAG:-pmc_train}_$i
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -- python3 $REPO/tools/train_step_r4.py --pre ${PRE:-0} --steps 2 --no-profile > $O.log 2>&1 || { echo "pass $i ($C) failed"; tail -3 $O.log; continue; }
  python3 $REPO/tools/pmc_dump.py $O "" | python3 -c "
import sys, re
flt = re.compile('${FILTER:-.}')
show = False
for line in sys.stdin:
    if not line.startswith(' '):
        show = bool(flt.search(line))
    if show:
        print(line, end='')
"
done
