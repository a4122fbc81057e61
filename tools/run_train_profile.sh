#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the training step at pre_size 0 and 100 -> profiles/<tag>_train.md
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${TAG}_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in 0 100; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pre$P -- python3 $REPO/tools/train_step_r4.py --pre $P --steps 4 --no-profile > $OUT/pre$P.log 2> $OUT/pre$P.err || { tail -5 $OUT/pre$P.err; exit 1; }
  python3 $REPO/tools/train_step_r4.py --pre $P --steps 3 > $OUT/phases$P.log 2>> $OUT/pre$P.err
  grep "^{" $OUT/phases$P.log >> $OUT/pre$P.log
done
cd $REPO && python3 tools/make_train_profile.py $TAG $OUT/pre0 $OUT/pre100 $OUT/pre0.log $OUT/pre100.log && cp profiles/${TAG}_train.md $OUT/
