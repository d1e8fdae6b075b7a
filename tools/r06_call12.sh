#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python3 -m pytest tests/test_gpu_solvers.py::test_solver_pos_att_all_channels_with_monitor "tests/test_gpu_solvers.py::test_solver_pos_att_channel_reference_grid" tests/test_gpu_uniwin.py -x -q -m gpu 2>&1 | tail -25
timeout 600 python3 tools/time_pos_att_run.py 2>&1 | tail -8
