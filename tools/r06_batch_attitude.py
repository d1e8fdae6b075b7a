"""Solver_attitude.simplified_run (3 x 1000x300x3, 5999 stages, float64): the channels as one launch per stage (hjb_solve_batch on the
table kernel) against three chains on threads of their own.  usage: python tools/r06_batch_attitude.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import hjbdp
hjbdp.Solver_attitude().simplified_run(n_stages=64)          # library load, first touch
for batched in (True, False, True, False):
    sa = hjbdp.Solver_attitude()
    sa.batch_channels = batched
    t0 = time.perf_counter()
    sa.simplified_run()
    wall = (time.perf_counter() - t0) * 1e3
    print("batched %-5s: simplified_run %.1f ms (channels together %.1f ms; groups %s; sweep ms per channel %s)"
          % (batched, wall, sa.wall_ms, sa.batch_groups, ["%.1f" % x for x in sa.sweep_ms]), flush=True)
