#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/ab_uniwin.sh uw_nest uw_nest_base uw_nest_pk uw_nest_m3 -- uw_nest uw_nest_base 2>&1 | tee gpurun_out/r06_ab_uniwin2.log
echo "== occupancy probe: K15 (uw_nest) with 6000 bytes of LDS padding = four workgroups per CU instead of five" | tee -a gpurun_out/r06_ab_uniwin2.log
for pad in 0 6000; do
  LDS_PAD=$pad HJBDP_LIB="$PWD/build/ab/uw_nest.so" timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage|grid" | sed "s/^.*'grid'/'grid'/" | cut -c1-160 | sed "s/^/pad=$pad: /" | tee -a gpurun_out/r06_ab_uniwin2.log
done
bash tools/r06_c4_pitch.sh 2>&1 | tee gpurun_out/r06_c4_pitch.log
