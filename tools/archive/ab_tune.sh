#!/bin/bash
# Run ON THE GPU BOX: MVM time / warm build time for a list of plx_tune settings.
#   TUNES="order_zcurve=1 order_zcurve=2" ELLS="1.0 0.25" EXTRA="--vd 12" tools/ab_tune.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for ELL in ${ELLS:-1.0 0.6931 0.5 0.4 0.25}; do
  for T in ${TUNES}; do
    echo "== ell=$ELL $T"
    timeout -k 10 120 python3 tools/prof_mvm.py --ell $ELL --tune ${T//,/ } --builds 3 $EXTRA 2>&1 | grep "apply" | cut -c1-${CUT:-100} || exit 1
  done
done
