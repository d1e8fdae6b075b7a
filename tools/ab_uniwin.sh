#!/bin/bash
# A/B timing of K15 builds (build/ab/<name>.so, tools/mkab.sh) on ONE box: 24^6 tabulated / on-the-fly model, then C3 for the names after "--"
# usage: bash tools/ab_uniwin.sh a b c -- x y      (list build/obj/ and build/san/ instead of build/ in .gpurunignore for the call)
cd "$GRAFT_REPO_ROOT" || exit 1
six=(); c3=(); seen=0
for v in "$@"; do if [ "$v" = "--" ]; then seen=1; elif [ $seen = 0 ]; then six+=("$v"); else c3+=("$v"); fi; done
for rep in 1 2; do
for v in "${six[@]}"; do
  HJBDP_LIB="$PWD/build/ab/$v.so" timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage" | sed "s/^/$v 6D tab: /"
  MODEL=1 HJBDP_LIB="$PWD/build/ab/$v.so" timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage" | sed "s/^/$v 6D model: /"
done
done
for v in "${c3[@]}"; do
  HJBDP_LIB="$PWD/build/ab/$v.so" timeout 600 python3 tools/time_c3.py 51 11 2 2>&1 | grep -E "stage [01]" | sed "s/^/$v C3: /"
done
