#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
L=gpurun_out/r06_call4.log; : > $L
timeout 600 python3 -m pytest tests/test_gpu_uniwin.py -x -q -m gpu 2>&1 | tail -5 >> $L
for rep in 1 2; do
for b in 256 64; do
  UW_BLOCK=$b timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage|packed2_mode" | sed "s/^.*grid'/grid'/" | sed "s/^/block=$b 6D tab: /" >> $L
  MODEL=1 UW_BLOCK=$b timeout 300 python3 tools/time_6d.py 24 11 3 2>&1 | grep -E "ms/stage" | sed "s/^/block=$b 6D model: /" >> $L
done
done
for b in 256 64; do
  UW_BLOCK=$b timeout 600 python3 tools/time_c3.py 51 11 2 2>&1 | grep -E "stage [01]|grid" | sed "s/^.*'block'/'block'/" | sed "s/^/block=$b C3: /" >> $L
done
O=gpurun_out/r06_call4_pmc; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python3 tools/time_6d.py 24 11 2 > $O/log_a 2>&1
python3 tools/pmc_summary.py $O/a k_backup_uniwin | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items(): print(k[-40:], {n:round(x['mean_per_launch']/1e9,4) for n,x in c.items()})" >> $L
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -- python3 tools/time_6d.py 24 11 2 > $O/log_b 2>&1
python3 tools/pmc_summary.py $O/b k_backup_uniwin | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,c in d.items(): print(k[-40:], {n:round(x['mean_per_launch']/1e9,4) for n,x in c.items()})" >> $L
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
timeout 900 python3 -m pytest "tests/test_gpu_deep.py::test_c3_full_size_second_stage_whole_grid_and_as_eight_slabs" tests/test_gpu_parity.py::test_two_rank_c3_bench_matches_single_rank -x -q -m gpu --durations=5 2>&1 | tail -25 >> $L
cat $L
