#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for n in 60 90 120 150; do
ORDER=0,2,3,1 timeout 300 python3 tools/time_posatt.py $n 10 7 2>&1 | grep "ran 7" | sed "s/^/n=$n xtwv: /"
done
