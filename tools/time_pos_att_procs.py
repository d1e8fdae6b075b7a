"""The four pos-att channels from four PROCESSES (one HIP context each) instead of four threads of one: does the device run more than two
sweeps at once?  usage: python tools/time_pos_att_procs.py [n_procs=4]"""
import os, sys, time, multiprocessing as mp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))

def work(i, bar, q):
    import hjbdp
    pa = hjbdp.Solver_pos_att()
    pa.cost_mode, pa.axis_order = "terms", "auto"
    sx, sv, st, sw = pa.grids()
    spec = pa._relabel(pa.build_channel_spec(sx, sv, st[i % 3], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)[0])[0]
    kw = dict(monitor_period=pa.monitor_period, monitor_tol=pa.monitor_tol, monitor_single=pa.monitor_single)
    bk = hjbdp.Backup(spec)
    bk.solve(100, **kw)                       # warm
    res = []
    for rep in range(3):
        bar.wait()
        t0 = time.perf_counter()
        out = bk.solve(pa.N_stage - 1, **kw)
        res.append((time.perf_counter() - t0) * 1e3)
    bk.close()
    q.put((i, res))

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ctx = mp.get_context("spawn")
    bar, q = ctx.Barrier(n), ctx.Queue()
    ps = [ctx.Process(target=work, args=(i, bar, q)) for i in range(n)]
    for p in ps: p.start()
    got = sorted(q.get() for _ in ps)
    for p in ps: p.join()
    for i, res in got:
        print("process %d: solve ms per repetition (all %d start together): %s" % (i, n, "  ".join("%.1f" % r for r in res)), flush=True)
