#!/bin/bash
# Run ON THE GPU BOX: MVM time and warm build time with first-touch (0) and Morton (2) vertex numbering.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for ELL in ${ELLS:-1.0 0.6931 0.5 0.4 0.35 0.3 0.25}; do
  for VO in 0 2; do
    echo "== ell=$ELL vertex_order=$VO"
    timeout -k 10 120 python3 tools/prof_mvm.py --ell $ELL --tune vertex_order=$VO --builds 3 $EXTRA 2>&1 | grep "apply" | cut -c1-90 || exit 1
  done
done
