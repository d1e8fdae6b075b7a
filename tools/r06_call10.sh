#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
L=gpurun_out/r06_tiles.log; : > $L
for t in 0 155 146 211 148 83 139; do
  UW_TILE=$t timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage" | sed "s/^/tile=$t 6D tab: /" >> $L
done
for t in 0 155 146 211 148 83 139; do
  UW_TILE=$t timeout 600 python3 tools/time_c3.py 51 11 2 2>&1 | grep -E "stage 1" | sed "s/^/tile=$t C3: /" >> $L
done
for b in 64; do
  UW_BLOCK=$b timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage" | sed "s/^/block=$b 6D tab: /" >> $L
  UW_BLOCK=$b timeout 600 python3 tools/time_c3.py 51 11 2 2>&1 | grep -E "stage 1" | sed "s/^/block=$b C3: /" >> $L
done
cat $L
