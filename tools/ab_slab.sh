#!/bin/bash
# A/B timing of a middle rank's slab stage (C4 sharded over N GPUs, one GPU here) for build/ab/<name>.so libraries:
#   tools/ab_slab.sh base prime      -> per library: N = 8 and N = 4, fused launch and interior + strips (hjb_rank_stage)
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in "$@"; do
  for N in 8 4; do
    HJBDP_LIB="$PWD/build/ab/$v.so" timeout 300 python3 tools/time_rank_slab.py $N 2>&1 | grep "fused\|hjb_rank_stage\|parts per" | sed "s/^/$v: /"
  done
done
done
