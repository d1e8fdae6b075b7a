#!/bin/bash
# The automatic stage-kernel choice on the small pos-att grids of tools/small_grid_variants.sh, both axis orders
cd "$GRAFT_REPO_ROOT"
for cfg in "A=1" "N_X=40 N_V=40 N_T=30 N_W=20" "N_X=60 N_V=60 N_T=40 N_W=30" "N_X=33 N_V=64 N_T=48 N_W=32" "N_X=96 N_V=40 N_T=30 N_W=20" "N_X=80 N_V=80 N_T=60 N_W=40"; do
  env $cfg python3 tools/time_posatt.py 0 400 -1 2>&1 | grep -E "ms/stage|refused" | sed 's/(halo.*//'
  env $cfg ORDER=0,2,3,1 python3 tools/time_posatt.py 0 400 -1 2>&1 | grep -E "ms/stage|refused|variant 7:" | sed 's/(halo.*//'
done
