#!/bin/bash
# round 6: the round's profile run (bench line + rocprofv3 kernel stats of the same command + VALU-rate calibration), the reference-sized
# configurations through the mirrors' defaults, and Solver_pos_att.simplified_run under every cost / axis-order setting
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/profile_round.sh 2>&1 | tail -25
cd "$GRAFT_REPO_ROOT"
HJB_MEASURE_6D=1 timeout 900 python3 tools/measure_configs.py > gpurun_out/round/measure_configs.txt 2>&1; tail -22 gpurun_out/round/measure_configs.txt
timeout 600 python3 tools/time_pos_att_run.py > gpurun_out/round/pos_att_run.log 2>&1; cat gpurun_out/round/pos_att_run.log
