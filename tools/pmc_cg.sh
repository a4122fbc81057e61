#!/bin/bash
# Run ON THE GPU BOX: HBM traffic counters (one per pass, kernel trace only) of the config-3 CG loop.
# FETCH_SIZE is in 64-byte units on gfx950 with the 32B/64B split (see MI355X_MICROARCH.md), so bytes are reported by
# tools/pmc_kernel.py as raw counts; DESIGN.md 4 applies the guide's corrections.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in ${COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  O=$REPO/gpurun_out/pmc_cg_$C
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -- python3 $REPO/tools/prof_cg.py --iters 10 "$@" > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
  python3 $REPO/tools/pmc_kernel.py $O
done
