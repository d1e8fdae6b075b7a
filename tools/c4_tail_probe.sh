#!/bin/bash
# Variant 7 on pos-att grids of several shapes x parts a column is swept in (cs_split): does a finer launch pay on the whole grid?
cd "$GRAFT_REPO_ROOT"
export ORDER=0,2,3,1
for cfg in "A=1" "F16=1" "N_T=128 N_W=100" "N_T=100 N_W=128" "N_T=160 N_W=80" "N_V=60" "N_V=200 N_T=80 N_W=80" "N_X=64 N_V=240"; do
  for sp in 1 2 3; do
    for rep in 1 2; do echo -n "$cfg split $sp: "; env $cfg CS_SPLIT=$sp python3 tools/time_posatt.py 120 60 7 2>&1 | grep -E "ms/stage" | sed 's/(halo.*//'; done
  done
done
