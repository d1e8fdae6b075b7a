#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mfma_table" --timeout 600 > $O/pytest_mfma.log 2>&1; echo "pytest rc=$?"; tail -n 12 $O/pytest_mfma.log
timeout 600 python3 tools/time_prep.py > $O/time_prep.jsonl 2> $O/time_prep.err; cat $O/time_prep.jsonl; tail -3 $O/time_prep.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prep_stats -- python3 tools/time_prep.py > $O/prep_stats.log 2>&1
find $O/prep_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/prep_kernel_stats.csv; grep -i "prep" $O/prep_kernel_stats.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
