#!/bin/bash
# round 6: counters of K15 (kernels_uniwin.h) against K3's window mode on the 24^6 grid (tabulated next angles), and K15's
# fabric traffic on C3 for two tilings of the chunk walk.  usage: bash tools/r06_uniwin_pmc.sh [c3]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_uw_pmc; rm -rf $O; mkdir -p $O
show() { python3 tools/pmc_summary.py "$1" "$2" | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items(): print(k[-40:], {n:round(x['mean_per_launch']/1e9,4) for n,x in c.items()})"; }
for u in 0 1; do
  K=$([ $u = 1 ] && echo k_backup_uniwin || echo k_backup_packed2)
  echo "== 24^6 tabulated, uniwin=$u"
  UNIWIN=$u timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/a$u -- python3 tools/time_6d.py 24 11 2 > $O/log_a$u 2>&1
  show $O/a$u $K
  UNIWIN=$u timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b$u -- python3 tools/time_6d.py 24 11 2 > $O/log_b$u 2>&1
  show $O/b$u $K
  UNIWIN=$u timeout 600 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/c$u -- python3 tools/time_6d.py 24 11 2 > $O/log_c$u 2>&1
  show $O/c$u $K
done
if [ "$1" = c3 ]; then
  for t in 0 211 83; do     # default 8x4x4; 8x4x4 spelt out (3 + 8*2 + 64*2 = 147 is the default; 211 = 3,2,3: 8x4x8; 83 = 3,2,1: 8x4x2)
    echo "== C3, uw_tile=$t"
    UW_TILE=$t timeout 900 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/c3_$t -- python3 tools/time_c3.py 51 11 2 > $O/log_c3_$t 2>&1
    grep -E "stage [01]:" $O/log_c3_$t
    show $O/c3_$t k_backup_uniwin
  done
fi
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
