"""The table kernel's launch size on a LARGE grid (C4 in the reference's own axis order, kernel variant 5 forced).  usage: python tools/r06_grid_sweep_k7.py [n=120]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import hjbdp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
pa = hjbdp.Solver_pos_att()
pa.cost_mode, pa.axis_order = "terms", None
pa.n_mesh_x = pa.n_mesh_v = pa.n_mesh_t = pa.n_mesh_w = n
sx, sv, st, sw = pa.grids()
spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
with hjbdp.Backup(spec, variant=5) as bk:
    blocks = -(-spec.nS // 256)
    print("variant", bk.info()["kernel_variant"], "states", spec.nS, "blocks", blocks, "automatic launch", bk.get_option("grid"), flush=True)
    for g in (0, 2048, 4096, 16384, 65536, blocks):
        if g:
            bk.set_option("grid", g)
        best = min(bk.solve(3)["sweep_ms"] for _ in range(2))
        print("   grid %7s: %.3f ms per stage" % (g or "auto", best / 3), flush=True)
