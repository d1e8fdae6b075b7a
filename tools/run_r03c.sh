cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c; mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
tail -c 600 $O/bench.err
python tools/time_rank_slab.py 8 > $O/rank_slab8.log 2>&1
tail -4 $O/rank_slab8.log
cat $O/bench.time
