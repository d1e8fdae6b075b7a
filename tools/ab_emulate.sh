#!/bin/bash
# emulated strong scaling (tools/emulate_ranks.py) for build/ab/<name>.so libraries on one box
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  echo "=== $v"
  HJBDP_LIB="$PWD/build/ab/$v.so" timeout 900 python3 tools/emulate_ranks.py 200 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
done
