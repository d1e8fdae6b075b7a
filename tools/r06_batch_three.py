"""The batch of three pos-att channels (x, z, failure: one group axis) WITHOUT channel y's chain beside it, and y alone: what one launch
for all four could reach.  usage: python tools/r06_batch_three.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import hjbdp
from hjbdp import core
for cost in ("f64", "terms"):
    pa = hjbdp.Solver_pos_att(); pa.cost_mode = cost
    sx, sv, st, sw = pa.grids()
    jobs = [(st[0], pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2),
            (st[1], pa.F_Thr2, pa.F_Thr3, pa.F_Thr8, pa.F_Thr9, pa.Qx2, pa.Qv2, pa.Qt2, pa.Qw2, pa.R2, pa.J3),
            (st[2], pa.F_Thr4, pa.F_Thr5, pa.F_Thr10, pa.F_Thr11, pa.Qx3, pa.Qv3, pa.Qt3, pa.Qw3, pa.R3, pa.J1),
            (st[0], [0.0], pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)]
    specs = [pa._relabel(pa.build_channel_spec(sx, sv, j[0], sw, *j[1:])[0])[0] for j in jobs]
    for name, sel in (("x, z, failure (one launch per stage)", [0, 2, 3]), ("y alone", [1]), ("all four (batch of three + y beside it)", [0, 1, 2, 3])):
        for S in (None, 2, 3):
            best = None
            for rep in range(3):
                outs, wall, variants, sizes = core.solve_batch([specs[i] for i in sel], 1999, monitor_period=50, monitor_tol=1e-2, monitor_single=True, cs_split=S)
                sw_ms = max(o["sweep_ms"] for o in outs)
                best = sw_ms if best is None else min(best, sw_ms)
            print("cost %-5s %-42s split %-4s: stage loop %.1f ms (groups %s)" % (cost, name, S, best, sizes), flush=True)
