#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py.
# Usage: tools/run_profiles.sh <tag>   -> gpurun_out/prof_<tag>/..., profiles/<tag>_summary.md, <tag>_pmc.json, <tag>_cg.md
# Counters are collected in their own runs, with --kernel-trace only (no sys/hip/hsa tracing), one
# counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass), and separately for the coarse
# (lengthscale 1.0, the bench headline) and fine (0.25) lattices so per-kernel averages do not mix regimes.
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 2 --skip-cpu-baseline --skip-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
for ELL in 1.0 0.25; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$OUT/pmc_${ELL}_${C}
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $REPO/bench.py $ARGS --skip-fine --ell $ELL > $D.json 2> $D.err
  done
done
cd $REPO && python3 tools/summarize_profile.py $OUT profiles/${TAG}_summary.md > $OUT/summary.log 2>&1 || (tail -20 $OUT/summary.log; exit 1)
cp profiles/${TAG}_summary.md profiles/${TAG}_pmc.json $OUT/
# the CG iteration (BASELINE.json configs[2]): per-kernel table + FETCH_SIZE / WRITE_SIZE passes -> profiles/<tag>_cg.md
tools/prof_cg_stats.sh ${TAG}_cg > $REPO/gpurun_out/${TAG}_cg.txt 2>&1
tools/pmc_cg.sh > $REPO/gpurun_out/${TAG}_cg_pmc.txt 2>&1
python3 tools/make_cg_profile.py $TAG && cp profiles/${TAG}_cg.md $OUT/
head -60 profiles/${TAG}_summary.md
