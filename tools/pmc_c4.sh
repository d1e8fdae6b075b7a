#!/bin/bash
# PMC passes on the C4 stage kernel (k_backup_tabled<float,float,4>): what bounds it?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc_c4
rm -rf $O && mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum" \
           "FETCH_SIZE" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 tools/time_posatt.py 120 2 5 > $O/p$i.log 2>&1
  echo "== $set"; python3 tools/pmc_summary.py $O/p$i k_backup_tabled | grep -E '"[A-Z_a-z0-9]+": \{|mean_per_launch' | paste - - | sed 's/  */ /g'
done
find $O -name "*kernel_trace.csv" -delete
