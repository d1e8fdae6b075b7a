#!/bin/bash
# Per-stage time of C4 in both fast axis orders and with float16 cost-to-go storage (tools/time_posatt.py).
cd "$GRAFT_REPO_ROOT" || exit 1
ORDER=0,2,3,1 timeout 300 python3 tools/time_posatt.py 120 50 7 2>&1 | grep "ran 7" | sed "s/^/xtwv: /"
ORDER=0,2,1,3 timeout 300 python3 tools/time_posatt.py 120 50 7 2>&1 | grep "ran 7" | sed "s/^/xtvw: /"
ORDER=0,2,3,1 F16=1 timeout 300 python3 tools/time_posatt.py 120 50 7 2>&1 | grep "ran 7" | sed "s/^/xtwv f16: /"
