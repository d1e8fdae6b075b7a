#!/bin/bash
# Small pos-att grids (the reference's own default 30x30x20x15 and a few larger): stage kernel 5 / 6 / 7 and variant 7's parts per column
cd "$GRAFT_REPO_ROOT"
for cfg in "A=1" "N_X=40 N_V=40 N_T=30 N_W=20" "N_X=60 N_V=60 N_T=40 N_W=30" "N_X=33 N_V=64 N_T=48 N_W=32" "N_X=96 N_V=40 N_T=30 N_W=20" "N_X=80 N_V=80 N_T=60 N_W=40"; do
  echo "== $cfg (reference order)"
  env $cfg python3 tools/time_posatt.py 0 400 5 6 2>&1 | grep -E "ms/stage|refused" | sed 's/(halo.*//; s/(x,v,theta,w) //'
  echo "== $cfg (x,theta,w,v)"
  env $cfg ORDER=0,2,3,1 python3 tools/time_posatt.py 0 400 5 6 7 2>&1 | grep -E "ms/stage|refused" | sed 's/(halo.*//; s/(x,theta,w,v) //'
  for sp in 2 4 8; do env $cfg CS_SPLIT=$sp ORDER=0,2,3,1 python3 tools/time_posatt.py 0 400 7 2>&1 | grep -E "ms/stage" | sed "s/(halo.*//; s/(x,theta,w,v) //; s/^/split $sp: /"; done
done
