#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export ORDER=0,2,3,1
bash tools/pmc_kernel.sh r02k_pmc colsweep python3 tools/time_posatt.py 120 4 7 > gpurun_out/r02k_pmc.log 2>&1
tail -n 60 gpurun_out/r02k_pmc/summary.json
