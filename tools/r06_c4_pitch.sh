#!/bin/bash
# does K10's texture-addresser load follow the ROW PITCH of J?  The column sweep gathers 61-lane row pieces
# (244 bytes) that start at (row * n0 + 60 * chunk) * 4 bytes: with n0 = 120 rows start on 32-byte phases.  Same problem, same two
# waves per row, same wave-steps per stage, only the axis-0 length - hence the pitch - differs:
#   n0 = 96  -> 384-byte rows (128-byte multiple)      n0 = 104 -> 416 (32-byte multiple only)
#   n0 = 112 -> 448-byte rows (64-byte multiple)       n0 = 120 -> 480 (32-byte multiple only; the C4 grid)
# If the pitch mattered, 96 / 112 would run faster PER STAGE than 104 / 120 (every stage is the same number of wave-steps).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for nx in 96 104 112 120; do
  N_X=$nx ORDER=0,2,3,1 IDX=auto timeout 300 python3 tools/time_posatt.py 120 50 7 2>&1 | grep "ran 7" | sed "s/^/n0=$nx: /"
done
done
O=gpurun_out/r06_c4_pitch; rm -rf $O; mkdir -p $O
for nx in 112 120; do
  N_X=$nx ORDER=0,2,3,1 IDX=auto timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/n$nx -- python3 tools/time_posatt.py 120 6 7 > $O/log_$nx 2>&1
  python3 tools/pmc_summary.py $O/n$nx k_backup_colsweep > $O/raw_$nx.json
  echo "== n0=$nx"; python3 tools/pmc_derive.py $O/raw_$nx.json | grep -E "tcp_line|ta_busy|l1_hit|gpu_cycles|TCP_TCC_READ|SQ_INSTS_VMEM_RD\"" 
done
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
