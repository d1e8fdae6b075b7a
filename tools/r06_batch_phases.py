"""Where Solver_pos_att.simplified_run's wall time goes outside the stage loop.  usage: python tools/r06_batch_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import hjbdp
from hjbdp import core
for cost in ("f64", "terms"):
    for rep in range(4):
        pa = hjbdp.Solver_pos_att(); pa.cost_mode = cost
        t0 = time.perf_counter(); pa.simplified_run(); w = (time.perf_counter() - t0) * 1e3
        ph = core.solve_batch.last_phases_ms
        print("cost %-5s run %d: simplified_run %.1f ms; solve_batch: create %.1f  sweep %.1f  close %.1f" % (cost, rep, w, ph["create"], ph["sweep"], ph["close"]), flush=True)
