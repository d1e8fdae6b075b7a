#!/bin/bash
# A/B of the table kernel (K7): parity tests on the in-tree build, then the pos-att channel at the reference's grid and the
# whole simplified_run timed with each library in build/ab/ named on the command line.  usage: tools/ab_k7.sh head k7b
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_k7; mkdir -p $O; rm -f $O/*.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_types.py tests/test_gpu_solvers.py -x -q -m gpu -k "random_problems or types or tab or pos_att or cost or attitude or position" --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 $O/pytest.log
for rep in 1 2; do for L in "$@"; do
  HJBDP_LIB=$PWD/build/ab/$L.so IDX=auto timeout 300 python3 tools/time_posatt.py 0 1999 5 2>&1 | tail -1 | sed "s/^/$L: /" | tee -a $O/time.log
done; done
for L in "$@"; do
  echo "== $L" | tee -a $O/run.log
  HJBDP_LIB=$PWD/build/ab/$L.so timeout 600 python3 tools/time_pos_att_run.py 2>&1 | tail -6 | tee -a $O/run.log
done
