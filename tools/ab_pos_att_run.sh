cd "$GRAFT_REPO_ROOT"
for L in head nosync; do echo "== $L"; HJBDP_LIB=$PWD/build/ab/$L.so timeout 600 python3 tools/time_pos_att_run.py 2>&1 | tail -6; done
