#!/bin/bash
# Run ON THE GPU BOX: HBM traffic counters (one per pass, kernel trace only) of the preconditioned solve (tools/pcg_r4.py):
# the factor build's and the preconditioner's kernels -> profiles/<tag>_pcg.md
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_pcg
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/pcg_r4.py > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 $REPO/tools/pcg_r4.py > $OUT/$C.log 2>&1 || { tail -5 $OUT/$C.log; exit 1; }
done
cd $REPO
{
  echo "# rocprofv3 of the preconditioned solve ($TAG): tools/pcg_r4.py"
  echo
  echo "N = 1e6, d = 8, lengthscale 0.6931, rank-100 factor (3 builds), [y | 10 probes from N(0, P)], plain and preconditioned CG at"
  echo "20 and 50 iterations (3 repetitions each).  Kernel trace + stats, then one PMC counter per pass (kernel trace only);"
  echo "counter unit KB per launch, FETCH_SIZE reports half the bytes of a wide coalesced read on gfx950."
  echo
  echo '```'
  grep "^{" $OUT/trace.log
  echo
  python3 tools/kstats.py $OUT/trace 40 "pc"
  echo
  for C in FETCH_SIZE WRITE_SIZE; do python3 tools/pmc_kernel.py $OUT/$C | grep -E "pcg_|pchol_|splat_onehot|cg_step"; done
  echo '```'
} > profiles/${TAG}_pcg.md
cp profiles/${TAG}_pcg.md $OUT/
