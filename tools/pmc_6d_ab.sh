#!/bin/bash
# usage: bash tools/pmc_6d_ab.sh base p1 ...   (VALU counts per build/ab/<name>.so on the 24^6 grid)
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  echo "== $v"
  export HJBDP_LIB="$PWD/build/ab/$v.so"
  bash tools/pmc_6d_quick.sh 2>&1 | grep -v "^$"
done
