#!/bin/bash
# Variant 7 on C4 (120^4 x 9): column -> XCD assignment (cs_xcd_mod, cs_xcd_axis) x parts per column, with the automatic choice first
cd "$GRAFT_REPO_ROOT"
export ORDER=0,2,3,1
for rep in 1 2; do echo -n "auto: "; python3 tools/time_posatt.py 120 60 7 2>&1 | grep -E "ms/stage" | sed 's/(halo.*//; s/.*workgroups, //'; done
for ax in 0 1; do for mod in 1 2 3 4 5 8 15; do for sp in 2; do
  echo -n "xcd_axis $ax xcd_mod $mod split $sp: "; CS_XCD_AXIS=$ax CS_XCD_MOD=$mod CS_SPLIT=$sp python3 tools/time_posatt.py 120 60 7 2>&1 | grep -E "ms/stage|refused" | sed 's/(halo.*//; s/.*workgroups, //'
done; done; done
