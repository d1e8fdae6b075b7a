#!/bin/bash
# Build an A/B copy of the library with some stage families recompiled from the working tree (and extra hipcc flags), leaving
# the in-tree libhjbdp.so alone:  tools/mkab.sh NAME "stage_packed2 stage_packed2_f16 stage_packed2_f32" [-DFLAG ...]
# -> build/ab/NAME.so, picked up by HJBDP_LIB=... (hjbdp/core.py).  The other objects come from build/obj: the script first
# runs __graft_entry__.build() (content-keyed, so a current tree costs nothing) and takes the compiler flags from there,
# so an A/B library never mixes stale objects or drifts from the product's flags.
# To send it to the GPU box, list build/obj/ instead of build/ in .gpurunignore for that call (tools/ab_k3.sh, tools/ab_time.sh).
set -e
N=$1; UNITS=$2; shift 2
cd "$(dirname "$0")/.."
python3 -c "import __graft_entry__ as g; g.build()"
FLAGS=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIPCC_FLAGS))")
mkdir -p build/ab/$N
rm -f build/ab/$N/*.o
C=optimal-control-dynamic-programming_amd/csrc
for f in $UNITS; do
  UF=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.UNIT_FLAGS.get('$f.hip', [])))")       # the unit's own flags
  /opt/rocm/bin/hipcc $FLAGS $UF -Iinclude -I$C "$@" -c $C/$f.hip -o build/ab/$N/$f.o &
done
wait
OBJS=""
for u in $C/*.hip; do
  b=$(basename $u .hip); skip=0
  for f in $UNITS; do [ "$b" = "$f" ] && skip=1; done
  if [ $skip = 0 ]; then
    [ -f build/obj/$b.o ] || { echo "missing build/obj/$b.o" >&2; exit 1; }
    OBJS="$OBJS build/obj/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/ab/$N/*.o -o build/ab/$N.so
ls -la build/ab/$N.so
