#!/bin/bash
# lazy window fill (K3): counters on 24^6 (instructions, waits, fabric traffic) and the C3 stage time, head against lazy.
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/pmc_6d_counters.sh "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" head lazy
bash tools/pmc_6d_counters.sh "FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_MISS_sum TCC_REQ_sum" head lazy
for v in head lazy; do
  HJBDP_LIB="$PWD/build/ab/$v.so" timeout 600 python3 tools/time_c3.py 51 11 2 2>&1 | tail -3 | sed "s/^/$v C3: /"
done
