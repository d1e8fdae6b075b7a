#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
L=optimal-control-dynamic-programming_amd/hjbdp/libhjbdp.so
export CS_COOP=0
cp build/exp/lib12.so $L
ORDER=0,2,3,1 timeout 300 python3 tools/time_posatt.py 120 10 7 2>&1 | grep "ran 7" | sed "s/^/exp12 (20 of 30 loads) xtwv: /"
