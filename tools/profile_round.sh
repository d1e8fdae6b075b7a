#!/bin/bash
# Round profile run on the GPU box: bench line, rocprofv3 kernel stats of the same command, VALU-rate calibration.
# Outputs under gpurun_out/round (copied into profiles/ afterwards).  usage: bash tools/profile_round.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/round
rm -rf $O && mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && timeout 300 /tmp/valu_rate > $O/valu_rate.json 2> $O/valu_rate.err; echo "valu_rate rc=$?"
timeout 900 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -c 400 $O/bench_n1.json
# the same command under rocprofv3 --stats (its own counter passes and the CPU baseline off: they are not kernels of this process)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-pmc > $O/stats.log 2>&1; echo "stats rc=$?"
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -8 $O/kernel_stats.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
du -sh $O
