#!/bin/bash
# Round profile run on the GPU box: full GPU tests, bench, rocprofv3 kernel stats and the PMC passes.
# Outputs under gpurun_out/ (copied into profiles/ afterwards).  usage: bash tools/profile_round.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/round
rm -rf $O && mkdir -p $O
timeout 1500 python -u -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_SQ -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_SQ.log 2>&1
python3 tools/make_pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ $O/pmc_traffic.json > /dev/null 2>$O/pmc_make.err; tail -2 $O/pmc_make.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -5 $O/kernel_stats.csv
# do not ship the bulky raw traces back
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
du -sh $O
