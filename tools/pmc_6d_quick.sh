#!/bin/bash
# VALU / SALU / LDS instruction counts + cycles of K3 on the 24^6 grid for the in-tree library (one rocprofv3 --pmc pass).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc6d; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O -- python3 tools/time_6d.py 24 11 2 > $O/log 2>&1
python3 tools/pmc_summary.py $O k_backup_packed2 | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items(): print(k[-24:], {n:round(x['mean_per_launch']/1e9,3) for n,x in c.items()})"
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/b -- python3 tools/time_6d.py 24 11 2 > $O/logb 2>&1
python3 tools/pmc_summary.py $O/b k_backup_packed2 | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items(): print(k[-24:], {n:round(x['mean_per_launch']/1e9,3) for n,x in c.items()})"
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
