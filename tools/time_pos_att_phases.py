"""Where Solver_pos_att.simplified_run's wall time goes: per channel the handle's creation and the 1,999-stage solve, alone, one after the
other, and side by side from four host threads (what hjbdp.core.solve_many does).  usage: python tools/time_pos_att_phases.py [cost_mode] [axis_order]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import hjbdp

cost_mode = sys.argv[1] if len(sys.argv) > 1 else "terms"
axis_order = sys.argv[2] if len(sys.argv) > 2 else "auto"
pa = hjbdp.Solver_pos_att()
pa.cost_mode, pa.axis_order = cost_mode, (None if axis_order == "None" else axis_order)
sx, sv, st, sw = pa.grids()
args = [(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2),
        (sx, sv, st[1], sw, pa.F_Thr2, pa.F_Thr3, pa.F_Thr8, pa.F_Thr9, pa.Qx2, pa.Qv2, pa.Qt2, pa.Qw2, pa.R2, pa.J3),
        (sx, sv, st[2], sw, pa.F_Thr4, pa.F_Thr5, pa.F_Thr10, pa.F_Thr11, pa.Qx3, pa.Qv3, pa.Qt3, pa.Qw3, pa.R3, pa.J1),
        (sx, sv, st[0], sw, [0.0], pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)]
t0 = time.perf_counter()
specs = [pa._relabel(pa.build_channel_spec(*a)[0])[0] for a in args]
print("host: four channel specs built in %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
n_st = pa.N_stage - 1
kw = dict(monitor_period=pa.monitor_period, monitor_tol=pa.monitor_tol, monitor_single=pa.monitor_single)

_dummies = []
def one(i, log):
    t0 = time.perf_counter()
    if os.environ.get("DUMMY_STREAMS"):            # probe: extra streams created before this handle's own (shifts its hardware queue)
        import torch
        for _ in range(int(os.environ["DUMMY_STREAMS"])):
            _dummies.append(torch.cuda.Stream())
    bk = hjbdp.Backup(specs[i])
    if os.environ.get("GRAPH"):
        bk.set_option("graph", int(os.environ["GRAPH"]))
    if os.environ.get("CS_SPLIT"):
        bk.set_option("cs_split", int(os.environ["CS_SPLIT"]))
    t1 = time.perf_counter()
    out = bk.solve(n_st, **kw)
    t2 = time.perf_counter()
    bk.close()
    t3 = time.perf_counter()
    log[i] = (t0, t1, t2, t3, out["stages_done"], bk_variant(out))

def bk_variant(out):
    return out.get("kernel_variant", -1)

for rep in range(3):
    log = {}
    t0 = time.perf_counter()
    for i in range(4):
        one(i, log)
    seq = (time.perf_counter() - t0) * 1e3
    print("rep %d one after the other: %.1f ms; per channel create / solve / close ms: %s" % (
        rep, seq, "  ".join("%.1f / %.1f / %.1f (%d stages)" % ((b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3, n) for a, b, c, d, n, _ in (log[i] for i in range(4)))), flush=True)
for rep in range(3):
    log = {}
    t0 = time.perf_counter()
    th = [threading.Thread(target=one, args=(i, log)) for i in range(int(os.environ.get('N_CH', '4')))]
    for t in th: t.start()
    for t in th: t.join()
    tog = (time.perf_counter() - t0) * 1e3
    print("rep %d side by side: %.1f ms; per channel [start +ms] create / solve / close ms: %s" % (
        rep, tog, "  ".join("[+%.1f] %.1f / %.1f / %.1f" % ((a - t0) * 1e3, (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3) for a, b, c, d, n, _ in (log[i] for i in sorted(log)))), flush=True)
