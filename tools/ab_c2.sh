#!/bin/bash
# A/B timing of K3 builds on C2 (101^3 x 21^3) on ONE box: build/ab/<name>.so for every name given
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
for v in "$@"; do
  HJBDP_LIB="$PWD/build/ab/$v.so" timeout 300 python3 tools/time_c2.py 101 21 40 2>&1 | grep -E "ms/stage" | sed "s/^/$v C2: /"
done
done
