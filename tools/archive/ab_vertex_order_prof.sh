#!/bin/bash
# Run ON THE GPU BOX: per-kernel tables (rocprofv3 --stats) of the MVM loop with both vertex numberings.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for ELL in ${ELLS:-1.0 0.25}; do
  for VO in 0 2; do
    O=$REPO/gpurun_out/vo_prof_${ELL}_$VO
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $REPO/tools/prof_mvm.py --ell $ELL --builds 3 --tune vertex_order=$VO > $O.log 2>&1 || exit 1
    echo "== ell=$ELL vertex_order=$VO"; grep "apply" $O.log
    python3 $REPO/tools/prof_mvm.py --stats $O
  done
done
