"""Where the wall time of the small 'deep' GPU tests goes (round 6, the suite's 540 s budget): create / solve / oracle phases of the
graph-replay test's 12 x 11 problem, the Kirk fixture and Solver_attitude.simplified_run's 1000 x 300 grid."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (os.path.join(ROOT, "optimal-control-dynamic-programming_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import hjbdp
from hjbdp import _abi
from oracle import c_oracle
from problems import nested_problem, random_terminal

def tick(label, t0):
    print("  %-46s %8.2f s" % (label, time.time() - t0), flush=True)
    return time.time()

print("graph-replay problem (12 x 11 x 3, float64, 151 stages)")
spec = nested_problem(8, (12, 11), (3,), dtype=np.float64, spread=0.05)
term = random_terminal(spec, 4)
t = time.time()
bk = hjbdp.Backup(spec); t = tick("create", t)
for mon in (0, 40):
    a = bk.solve(151, terminal=term, monitor_period=mon, monitor_tol=0.0); t = tick("solve (first, monitor %d)" % mon, t)
    a = bk.solve(151, terminal=term, monitor_period=mon, monitor_tol=0.0); t = tick("solve (cached graph)", t)
bk.set_option("graph", 0); t = tick("set_option graph 0", t)
a = bk.solve(151, terminal=term); t = tick("solve eager", t)
bk.close(); t = tick("close", t)
r = c_oracle.sweep(_abi, spec, 151, terminal=term); t = tick("oracle sweep", t)

print("Kirk fixture (35 x 35 x 100, float64, 129 stages)")
ds = hjbdp.Dynamic_Solver(precision="double"); ds.N, ds.dx, ds.du = 130, 35, 100
spec = ds.build_spec()
t = time.time()
with hjbdp.Backup(spec) as bk:
    t = tick("create", t)
    o = bk.solve(129, keep_J=True, keep_idx=True); t = tick("solve keep_J keep_idx", t)
r = c_oracle.sweep(_abi, spec, 129, keep_J=True, keep_idx=True); t = tick("oracle sweep", t)

print("Solver_attitude.simplified_run 1000 x 300 x 3, float64, 200 stages")
sa = hjbdp.Solver_attitude()
t = time.time()
sa.simplified_run(n_stages=200); t = tick("mirror simplified_run (3 channels)", t)
spec, s_w, s_t = sa.build_spec_simplified(0); t = tick("build spec", t)
r = c_oracle.sweep(_abi, spec, 200, keep_J=True, keep_idx=True); t = tick("oracle sweep keep all", t)
r2 = c_oracle.sweep(_abi, spec, 200); t = tick("oracle sweep final only", t)
with hjbdp.Backup(spec) as bk:
    t = tick("create", t)
    o = bk.solve(200, keep_J=True, keep_idx=True); t = tick("solve keep_J keep_idx", t)
ok = np.array_equal(o["J_stages"], r["J_stages"]); t = tick("compare 480 MB", t)
