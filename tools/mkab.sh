#!/bin/bash
# Build an A/B copy of the library with some stage families recompiled from the working tree (and extra hipcc flags), leaving
# the in-tree libhjbdp.so alone:  tools/mkab.sh NAME "stage_packed2 stage_packed2_f16 stage_packed2_f32" [-DFLAG ...]
# -> build/ab/NAME.so, picked up by HJBDP_LIB=... (hjbdp/core.py).  Needs build/obj from a previous __graft_entry__.build().
# To send it to the GPU box, list build/obj/ instead of build/ in .gpurunignore for that call (tools/ab_k3.sh, tools/ab_time.sh).
set -e
N=$1; UNITS=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p build/ab/$N
C=optimal-control-dynamic-programming_amd/csrc
for f in $UNITS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$C "$@" -c $C/$f.hip -o build/ab/$N/$f.o &
done
wait
OBJS=""
for o in build/obj/*.o; do
  b=$(basename $o .o); skip=0
  for f in $UNITS; do [ "$b" = "$f" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/ab/$N/*.o -o build/ab/$N.so
ls -la build/ab/$N.so
