"""Solver_pos_att.simplified_run (reference grid, mirror defaults) with the batched channels' columns cut into S parts.  usage: python tools/r06_batch_split.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import hjbdp
from hjbdp import core
for cost in ("f64", "terms"):
    for S in (None, 2, 3, 4, 5):
        best = None
        for rep in range(3):
            pa = hjbdp.Solver_pos_att(); pa.cost_mode = cost
            sx, sv, st, sw = pa.grids()
            jobs = [(st[0], pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2),
                    (st[1], pa.F_Thr2, pa.F_Thr3, pa.F_Thr8, pa.F_Thr9, pa.Qx2, pa.Qv2, pa.Qt2, pa.Qw2, pa.R2, pa.J3),
                    (st[2], pa.F_Thr4, pa.F_Thr5, pa.F_Thr10, pa.F_Thr11, pa.Qx3, pa.Qv3, pa.Qt3, pa.Qw3, pa.R3, pa.J1),
                    (st[0], [0.0], pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)]
            specs = [pa._relabel(pa.build_channel_spec(sx, sv, j[0], sw, *j[1:])[0])[0] for j in jobs]
            t0 = time.perf_counter()
            outs, wall, variants, sizes = core.solve_batch(specs, 1999, monitor_period=50, monitor_tol=1e-2, monitor_single=True, cs_split=S)
            w = (time.perf_counter() - t0) * 1e3
            best = w if best is None else min(best, w)
        print("cost %-5s batched split %-4s: %.1f ms (groups %s; sweep ms per problem %s)" % (cost, S, best, sizes, ["%.1f" % o["sweep_ms"] for o in outs]), flush=True)
    pa = hjbdp.Solver_pos_att(); pa.cost_mode = cost; pa.batch_channels = False
    best = None
    for rep in range(3):
        t0 = time.perf_counter(); pa.simplified_run(); w = (time.perf_counter() - t0) * 1e3
        best = w if best is None else min(best, w)
    print("cost %-5s four chains on threads (no batching): %.1f ms" % (cost, best), flush=True)
