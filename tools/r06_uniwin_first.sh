#!/bin/bash
# round 6, first contact of K15 (kernels_uniwin.h) with the GPU: its parity tests, then A/B timing against K3's window modes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r06_uniwin_first.log
: > $L
timeout 900 python3 -m pytest tests/test_gpu_uniwin.py -x -q -m gpu 2>&1 | tail -30 >> $L
for rep in 1 2; do
for u in 0 1; do
  UNIWIN=$u timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage|packed2_mode" | sed "s/^/uniwin=$u 6D tab: /" >> $L
  MODEL=1 UNIWIN=$u timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage|packed2_mode" | sed "s/^/uniwin=$u 6D model: /" >> $L
done
done
for u in 0 1; do
  UNIWIN=$u timeout 600 python3 tools/time_c3.py 51 11 2 2>&1 | grep -E "stage|packed2_mode|range" | sed "s/^/uniwin=$u C3: /" >> $L
done
cat $L
