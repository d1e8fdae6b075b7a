"""Host-side set-up time of a C4 handle (spec build, hjb_create with its tables and plans), for the INTEGRATION notes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "optimal-control-dynamic-programming_amd"))
import hjbdp
pa = hjbdp.Solver_pos_att(); pa.cost_mode = "terms"
pa.n_mesh_x = pa.n_mesh_v = pa.n_mesh_t = pa.n_mesh_w = 120
sx, sv, st, sw = pa.grids()
spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
spec, _ = hjbdp.permute_state_axes(spec, (0, 2, 3, 1))
for rep in range(2):
    t0 = time.time()
    bk = hjbdp.Backup(spec)
    t1 = time.time()
    info = bk.info()
    out = bk.solve(1)
    t2 = time.time()
    print("create %.3f s, first solve(1) incl. J alloc + copies %.3f s, variant %d" % (t1 - t0, t2 - t1, info["kernel_variant"]))
    bk.close()
