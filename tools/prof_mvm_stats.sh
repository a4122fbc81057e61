#!/bin/bash
# Run ON THE GPU BOX: per-kernel table (rocprofv3 --stats) of tools/prof_mvm.py for each plx_tune setting in TUNES.
#   TUNES="order_zcurve=1 order_zcurve=2" ELL=1.0 EXTRA="--vd 1" tools/prof_mvm_stats.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for T in ${TUNES:-vertex_order=1}; do
  O=$REPO/gpurun_out/pm_${T//[=,]/_}
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $REPO/tools/prof_mvm.py --ell ${ELL:-1.0} --builds 3 $EXTRA --tune ${T//,/ } > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
  echo "== $T"; grep apply $O.log | cut -c1-80
  python3 $REPO/tools/prof_mvm.py --stats $O | grep -v "rocprim\|rocclr" | head -${ROWS:-12}
done
