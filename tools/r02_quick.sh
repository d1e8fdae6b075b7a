#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_flat_api.py -x -q --timeout 600 > $O/pytest_flat.log 2>&1; echo "pytest flat rc=$?"; tail -n 6 $O/pytest_flat.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "two_rank" --timeout 600 > $O/pytest_two.log 2>&1; echo "pytest two rc=$?"; tail -n 6 $O/pytest_two.log
bash tools/profile_round.sh
