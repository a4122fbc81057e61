cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_c4 -- python3 $GRAFT_REPO_ROOT/tools/prof_mvm.py --n 4000000 --vd 11 --reps 10 --tune block_multi=2 > $OUT/prof_c4.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_mvm.py --stats $OUT/prof_c4 > $OUT/prof_c4_stats.txt 2>&1
cat $OUT/prof_c4_stats.txt | head -30
tail -2 $OUT/prof_c4.log
