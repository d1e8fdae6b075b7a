#!/bin/bash
# A/B timing of K3 builds on the 6-D 24^6 grid (tabulated and on-the-fly model) on ONE box: build/ab/<name>.so for every name given
# usage: bash tools/ab_6d.sh base A B     (list build/obj/ instead of build/ in .gpurunignore for the call)
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in "$@"; do
  HJBDP_LIB="$PWD/build/ab/$v.so" timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage" | sed "s/^/$v 6D tab: /"
  MODEL=1 HJBDP_LIB="$PWD/build/ab/$v.so" timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage" | sed "s/^/$v 6D model: /"
done
done
