#!/bin/bash
# VALU instruction count + launch time of K3 for several builds (build/ab/<name>.so) on ONE box.  usage: bash tools/ab_k3_pmc.sh P RR
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  export HJBDP_LIB="$PWD/build/ab/$v.so"
  for w in "time_c2.py 101 21 3" "time_6d.py 24 11 2"; do
    O=gpurun_out/abpmc/$v-${w%% *}; rm -rf $O; mkdir -p $O
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -- python3 tools/$w > $O/log 2>&1
    python3 tools/pmc_summary.py $O k_backup_packed2 | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items(): print('$v', '$w', k[-22:], {n:round(x['mean_per_launch']/1e6,2) for n,x in c.items()})"
  done
done
find gpurun_out/abpmc -name "*.csv" -delete; find gpurun_out/abpmc -name "*.db" -delete
