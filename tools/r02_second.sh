#!/bin/bash
# round 2, second GPU call: VALU-rate calibration, variant-7 parity tests, C4 timings, PMC of the column-sweep kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02b; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && timeout 120 /tmp/valu_rate > $O/valu_rate.json 2> $O/valu_rate.err
echo "valu_rate rc=$?"; tail -n 4 $O/valu_rate.json
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "colsweep" --timeout 600 > $O/pytest_colsweep.log 2>&1; echo "pytest rc=$?"; tail -n 15 $O/pytest_colsweep.log
timeout 300 python3 tools/time_posatt.py 120 5 6 > $O/c4_v6.log 2>&1; cat $O/c4_v6.log
ORDER=0,2,1,3 timeout 300 python3 tools/time_posatt.py 120 10 7 6 > $O/c4_v7_xtvw.log 2>&1; cat $O/c4_v7_xtvw.log
ORDER=0,2,3,1 timeout 300 python3 tools/time_posatt.py 120 10 7 > $O/c4_v7_xtwv.log 2>&1; cat $O/c4_v7_xtwv.log
ORDER=0,2,1,3 F16=1 timeout 300 python3 tools/time_posatt.py 120 10 7 > $O/c5_v7.log 2>&1; cat $O/c5_v7.log
for t in 1,1 4,4 16,4 4,16 120,1 1,120; do ORDER=0,2,1,3 CS_TILE=$t timeout 200 python3 tools/time_posatt.py 120 6 7 2>&1 | sed "s/^/tile $t: /" | tee -a $O/c4_v7_tiles.log; done
ORDER=0,2,1,3 timeout 900 bash tools/pmc_kernel.sh r02b/pmc_colsweep k_backup_colsweep python3 tools/time_posatt.py 120 3 7 > $O/pmc_colsweep.log 2>&1
tail -n 60 $O/pmc_colsweep.log
