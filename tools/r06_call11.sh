#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python3 -m pytest tests -x -q -m gpu --durations=30 > gpurun_out/r06_suite_durations_4.log 2>&1; tail -3 gpurun_out/r06_suite_durations_4.log
bash tools/r06_final_profile.sh
