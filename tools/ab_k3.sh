#!/bin/bash
# A/B timing of K3 builds on ONE box: build/ab/<name>.so for every name given (HJBDP_LIB picks the library; the in-tree one is untouched)
# usage: bash tools/ab_k3.sh base A B
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in "$@"; do
  HJBDP_LIB="$PWD/build/ab/$v.so" timeout 300 python3 tools/time_c2.py 101 21 40 2>&1 | grep -E "ms/stage|sum J" | tr '\n' ' ' | sed "s/^/$v C2: /"; echo
  HJBDP_LIB="$PWD/build/ab/$v.so" timeout 300 python3 tools/time_6d.py 24 11 4 2>&1 | grep -E "ms/stage|ms per stage|sum" | tr '\n' ' ' | sed "s/^/$v 6D: /"; echo
done
done
