#!/bin/bash
# A/B timing of K10 builds on ONE box (the pool's boxes clock 2.15 - 2.21 GHz): build/ab/<name>.so for every name given
# (the loader takes the alternate build from HJBDP_LIB, hjbdp/core.py: the in-tree library is never overwritten)
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in "$@"; do
  HJBDP_LIB="$PWD/build/ab/$v.so" ORDER=0,2,3,1 IDX=auto timeout 300 python3 tools/time_posatt.py 120 50 7 2>&1 | grep "ran 7" | sed "s/^/$v: /"
done
done
