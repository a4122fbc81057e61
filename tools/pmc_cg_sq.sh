#!/bin/bash
# Run ON THE GPU BOX: wave-state and L1 counters of the CG loop's kernels, one counter group per pass (kernel trace only).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1)); O=$REPO/gpurun_out/pmc_sq_$i
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -- python3 $REPO/tools/prof_cg.py --iters 6 "$@" > $O.log 2>&1 || { tail -3 $O.log; continue; }
  python3 $REPO/tools/pmc_dump.py $O "${FILTER:-blur_pair}"
done
