#!/bin/bash
# PMC passes (rocprofv3 --pmc, one counter set per pass, --kernel-trace only) on one stage kernel.
# usage: bash tools/pmc_kernel.sh OUT_NAME KERNEL_FILTER python3 <script> [args...]
# Writes gpurun_out/OUT_NAME/summary.json: mean counter value per launch of the kernels matching KERNEL_FILTER,
# plus the derived quantities DESIGN.md quotes (tools/pmc_derive.py).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
NAME=$1; FILT=$2; shift 2
O=gpurun_out/$NAME
rm -rf $O && mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- "$@" > $O/p$i.log 2>&1
  echo "== pass $i rc=$? : $set"
done
python3 tools/pmc_summary.py $O "$FILT" > $O/raw.json
python3 tools/pmc_derive.py $O/raw.json > $O/summary.json
cat $O/summary.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
