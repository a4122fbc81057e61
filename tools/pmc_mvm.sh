#!/bin/bash
# Run ON THE GPU BOX: wave-state / L1 / L2 / TLB counters of the kernels of tools/prof_mvm.py, one counter group per pass.
#   EXTRA="--n 4000000 --vd 11 --tune perm_rows=0" FILTER="gather_in|slice_vec" tools/pmc_mvm.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); O=$REPO/gpurun_out/${TAG:-pmc_mvm}_$i
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -- python3 $REPO/tools/prof_mvm.py --reps 5 $EXTRA > $O.log 2>&1 || { echo "pass $i ($C) failed"; tail -3 $O.log; continue; }
  python3 $REPO/tools/pmc_dump.py $O "" | python3 -c "
import sys, re
flt = re.compile('${FILTER:-.}')
show = False
for line in sys.stdin:
    if not line.startswith(' '):
        show = bool(flt.search(line))
    if show:
        print(line, end='')
"
done
