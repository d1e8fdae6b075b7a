#!/bin/bash
# L2 / HBM counters of K3 on the 6-D grid for several builds (build/ab/<name>.so).  usage: bash tools/ab_k3_l2.sh CUR XW
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  export HJBDP_LIB="$PWD/build/ab/$v.so"
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE"; do
    O=gpurun_out/abl2/$v; rm -rf $O; mkdir -p $O
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O -- python3 tools/time_6d.py 24 11 2 > $O/log 2>&1
    python3 tools/pmc_summary.py $O k_backup_packed2 | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items(): print('$v', k[-22:], {n:round(x['mean_per_launch']/1e6,2) for n,x in c.items()})"
  done
done
find gpurun_out/abl2 -name "*.csv" -delete; find gpurun_out/abl2 -name "*.db" -delete
