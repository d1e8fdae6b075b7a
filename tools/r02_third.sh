#!/bin/bash
# round 2, third GPU call: variant 7 v3 (DPP form) parity + timing + PMC; flat API / probe tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_flat_api.py -x -q -k "colsweep or flat or probe or progress" --timeout 600 > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -n 25 $O/pytest_sel.log
ORDER=0,2,1,3 timeout 300 python3 tools/time_posatt.py 120 10 7 > $O/c4_v7_xtvw.log 2>&1; cat $O/c4_v7_xtvw.log
ORDER=0,2,3,1 timeout 300 python3 tools/time_posatt.py 120 10 7 > $O/c4_v7_xtwv.log 2>&1; cat $O/c4_v7_xtwv.log
ORDER=0,2,1,3 F16=1 timeout 300 python3 tools/time_posatt.py 120 10 7 > $O/c5_v7.log 2>&1; cat $O/c5_v7.log
ORDER=0,2,1,3 timeout 900 bash tools/pmc_kernel.sh r02c/pmc_colsweep k_backup_colsweep python3 tools/time_posatt.py 120 3 7 > $O/pmc_colsweep.log 2>&1
tail -n 32 $O/pmc_colsweep.log
