#!/bin/bash
# K7 with the next control's table entries requested one control ahead: timing of build/ab libraries on one box, then the parity slice.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_k7_pf; mkdir -p $O; rm -f $O/*.log
for rep in 1 2; do for L in "$@"; do
  HJBDP_LIB=$PWD/build/ab/$L.so timeout 300 python3 tools/time_posatt.py 0 1999 5 2>&1 | tail -1 | sed "s/^/$L: /" | tee -a $O/time.log
done; done
for L in "$@"; do
  HJBDP_LIB=$PWD/build/ab/$L.so N_X=60 N_V=60 N_T=40 N_W=30 timeout 300 python3 tools/time_posatt.py 0 200 5 2>&1 | tail -1 | sed "s/^/$L: /" | tee -a $O/time.log
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_types.py tests/test_gpu_solvers.py tests/test_gpu_flat_api.py -x -q -m gpu -k "random_problems or types or tab or pos_att or cost or attitude or position or probe or policy or kirk or dynamic" --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 $O/pytest.log
