#!/bin/bash
# Run ON THE GPU BOX: per-kernel table of the config-3 CG loop (tools/prof_cg.py) under rocprofv3 --stats.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-cg}
shift
cd /tmp && export TMPDIR=/tmp
O=$REPO/gpurun_out/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $REPO/tools/prof_cg.py "$@" > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
grep trial $O.log
python3 $REPO/tools/prof_mvm.py --stats $O
