#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py.
# Usage: tools/run_profiles.sh <tag>     -> gpurun_out/prof_<tag>/{trace,fetch,write}/...
# Counters are collected in their own runs, with --kernel-trace only (no sys/hip/hsa tracing).
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 2 --skip-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $REPO/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err
cd $REPO && python3 tools/summarize_profile.py $OUT profiles/${TAG}_summary.md > $OUT/summary.log 2>&1 || (tail -20 $OUT/summary.log; exit 1)
cp profiles/${TAG}_summary.md $OUT/
tail -60 profiles/${TAG}_summary.md
