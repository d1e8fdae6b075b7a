#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
L=gpurun_out/r06_claim.log; : > $L
timeout 600 python3 -m pytest tests/test_gpu_uniwin.py -x -q -m gpu 2>&1 | tail -3 >> $L
for rep in 1 2; do
for c in 0 1; do
  UW_CLAIM=$c timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage" | sed "s/^/claim=$c 6D tab: /" >> $L
  MODEL=1 UW_CLAIM=$c timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage" | sed "s/^/claim=$c 6D model: /" >> $L
done
done
for c in 0 1; do
  UW_CLAIM=$c timeout 600 python3 tools/time_c3.py 51 11 2 2>&1 | grep -E "stage [01]" | sed "s/^/claim=$c C3: /" >> $L
done
O=gpurun_out/r06_claim_pmc; rm -rf $O; mkdir -p $O
for c in 1 0; do
  UW_CLAIM=$c timeout 400 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/c$c -- python3 tools/time_c3.py 51 11 1 > $O/log_c$c 2>&1
  echo "== C3 one stage, claim=$c (KiB for the SIZE counters)" >> $L
  python3 tools/pmc_summary.py $O/c$c k_backup_uniwin | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items(): print(k[-40:], {n:round(x['mean_per_launch']/1e9,4) for n,x in c.items()})" >> $L
done
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
timeout 600 python3 tools/time_pos_att_phases.py f64 auto 2>&1 | tail -8 >> $L
timeout 900 python3 tools/emulate_ranks.py 4 c3 2>&1 | tail -12 >> $L
cat $L
