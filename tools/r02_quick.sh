#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02f; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && timeout 200 /tmp/valu_rate > $O/valu_rate.json 2> $O/valu_rate.err; echo "valu_rate rc=$?"
timeout 900 python -m pytest tests/test_gpu_flat_api.py -x -q --timeout 600 > $O/pytest_flat.log 2>&1; echo "pytest flat rc=$?"; tail -n 6 $O/pytest_flat.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "two_rank" --timeout 600 > $O/pytest_two.log 2>&1; echo "pytest two rc=$?"; tail -n 6 $O/pytest_two.log
timeout 900 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -c 3000 $O/bench_n1.json; tail -n 5 $O/bench_n1.err
