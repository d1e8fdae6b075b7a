#!/bin/bash
# Quick look at the pos-att stage kernel on the GPU box: both axis orders timed, then the column-sweep / multi-GPU parity tests.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/quick; mkdir -p $O; rm -f $O/quick.log
ORDER=0,2,3,1 timeout 300 python3 tools/time_posatt.py 120 50 7 2>&1 | tail -2 | sed "s/^/xtwv: /" | tee -a $O/quick.log
ORDER=0,2,1,3 timeout 300 python3 tools/time_posatt.py 120 50 7 2>&1 | tail -1 | sed "s/^/xtvw: /" | tee -a $O/quick.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_flat_api.py -x -q -k "colsweep or multi or two_rank" --timeout 600 > $O/pytest_cs.log 2>&1; echo "pytest rc=$?"; tail -n 4 $O/pytest_cs.log
timeout 900 python -m pytest tests/test_gpu_solvers.py -x -q -k "c4" --timeout 800 > $O/pytest_c4.log 2>&1; echo "pytest c4 rc=$?"; tail -n 3 $O/pytest_c4.log
