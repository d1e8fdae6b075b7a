#!/bin/bash
# K3 A/B on one box: 24^6 timing of the named build/ab libraries, then the K3 parity slice on the in-tree library.
# usage: bash tools/ab_k3_pair.sh head pair     (list build/obj/ instead of build/ in .gpurunignore for the call)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_k3_pair; mkdir -p $O; rm -f $O/*.log
bash tools/ab_6d.sh "$@" 2>&1 | tee $O/time.log
timeout 1500 python -m pytest tests/test_gpu_deep.py tests/test_gpu_solvers.py tests/test_gpu_parity.py -x -q -m gpu -k "6d or attitude or c3 or packed or nested or window or slab" --timeout 1200 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $O/pytest.log
