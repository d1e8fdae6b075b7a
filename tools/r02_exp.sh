#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
L=optimal-control-dynamic-programming_amd/hjbdp/libhjbdp.so
for e in 25 22; do
cp build/exp/lib$e.so $L
ORDER=0,2,3,1 timeout 300 python3 tools/time_posatt.py 120 10 7 2>&1 | grep "ran 7" | sed "s/^/exp$e xtwv: /"
done
