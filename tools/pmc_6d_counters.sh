#!/bin/bash
# Any counter set on K3's 24^6 stage for build/ab/<name>.so:  bash tools/pmc_6d_counters.sh "<counters>" name [name ...]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
CTRS=$1; shift
for v in "$@"; do
  export HJBDP_LIB="$PWD/build/ab/$v.so"
  O=gpurun_out/pmc6d_$v; rm -rf $O; mkdir -p $O
  timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $O -- python3 tools/time_6d.py 24 11 2 > $O/log 2>&1
  echo "== $v"
  python3 tools/pmc_summary.py $O k_backup_packed2 | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items(): print(k[-24:], {n:round(x['mean_per_launch']/1e9,4) for n,x in c.items()})"
  find $O -name "*.csv" -delete; find $O -name "*.db" -delete
done
