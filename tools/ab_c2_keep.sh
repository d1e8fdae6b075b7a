#!/bin/bash
# C2 A/B (K3 modes 1 / 4) on one box: timing of the named build/ab libraries, then the C2-mode parity slice on the in-tree library.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_c2_keep; mkdir -p $O; rm -f $O/*.log
bash tools/ab_c2.sh "$@" 2>&1 | tee $O/time.log
timeout 1500 python -m pytest tests/test_gpu_deep.py tests/test_gpu_solvers.py tests/test_gpu_parity.py tests/test_gpu_types.py -x -q -m gpu -k "c2 or position or packed or nested or f16 or slab or axis0" --timeout 1200 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $O/pytest.log
