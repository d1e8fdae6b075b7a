#!/bin/bash
# round 2, first GPU call: VALU-rate calibration, DPP probe, C4 baseline + PMC of the lean row kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02a; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate > $O/valu_rate.json 2> $O/valu_rate.err
tail -n 12 $O/valu_rate.json
python3 tools/time_posatt.py 120 5 6 > $O/c4_baseline.log 2>&1; cat $O/c4_baseline.log
bash tools/pmc_kernel.sh r02a/pmc_rowlean k_backup_rowlean python3 tools/time_posatt.py 120 2 6 > $O/pmc_rowlean.log 2>&1
tail -n 40 $O/pmc_rowlean.log
