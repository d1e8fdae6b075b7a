#!/usr/bin/env python3
"""Times the reference-sized configurations (and the C4 synthetic grid) on one
MI355X through the host mirrors; prints a markdown table.  Not a test, not the
bench contract - it feeds the numbers quoted in DESIGN.md / README.md."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "optimal-control-dynamic-programming_amd"))
import numpy as np
import hjbdp
from hjbdp.matlab_compat import sym_linspace_pos_att

rows = []


def add(name, backups, ms, variant, note=""):
    rows.append((name, backups, ms, backups / (ms * 1e-3), variant, note))


def solve_timed(spec, n_st, **kw):
    with hjbdp.Backup(spec) as bk:
        bk.solve(min(n_st, 3))
        out = bk.solve(n_st, **kw)
        v = bk.info()["kernel_variant"]
    return out, v


ds = hjbdp.Dynamic_Solver(precision="double"); ds.N, ds.dx, ds.du = 130, 35, 100
out, v = solve_timed(ds.build_spec(), 129, keep_J=True, keep_idx=True)
add("C1a Kirk fixture 35x35x100, 129 stages, f64", 35 * 35 * 100 * 129, out["sweep_ms"], v)
ds = hjbdp.Dynamic_Solver()
out, v = solve_timed(ds.build_spec(), 199, keep_J=True, keep_idx=True)
add("C1b Kirk defaults 100x100x1000, 199 stages, f32", 100 * 100 * 1000 * 199, out["sweep_ms"], v)
sp = hjbdp.Solver_position()
spec, _, _ = sp.build_spec(0)
out, v = solve_timed(spec, 5999)
add("Solver_position channel 201x201x3, 5999 stages, f64", spec.nS * 3 * 5999, out["sweep_ms"], v, "launch-bound")
sa = hjbdp.Solver_attitude()
spec, _, _ = sa.build_spec_simplified(0)
out, v = solve_timed(spec, 5999)
add("Solver_attitude simplified channel 1000x300x3, 5999 stages, f64", spec.nS * 3 * 5999, out["sweep_ms"], v)
sa = hjbdp.Solver_attitude(n_mesh_w=11, n_mesh_q=10)
spec = sa.build_spec_full()
out, v = solve_timed(spec, 19)
add("Solver_attitude.run 11^3x10^3 x 27, 19 stages, f32 (reference axis order)", spec.nS * 27 * 19, out["sweep_ms"], v)
pspec, _ = hjbdp.permute_state_axes(spec, sa.AXIS_ORDER)
out, v = solve_timed(pspec, 19)
add("Solver_attitude.run, same, axes relabelled (angles first, w3 last)", spec.nS * 27 * 19, out["sweep_ms"], v)
pa = hjbdp.Solver_pos_att()
sx, sv, st, sw = pa.grids()
spec_ref, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, 6, 6, .5, .5, .1, pa.J2)
spec, _ = pa._relabel(spec_ref)          # the mirror's defaults: axis_order "auto", cost_mode 'f64'
out, v = solve_timed(spec, 1999, monitor_period=50, monitor_tol=1e-2)
add("Solver_pos_att channel 30x30x20x15 x 9, <=1999 stages (monitor), f32, the mirror's defaults", spec.nS * 9 * out["stages_done"],
    out["sweep_ms"], v, "stopped after %d stages" % out["stages_done"])
out, v = solve_timed(spec_ref, 1999, monitor_period=50, monitor_tol=1e-2)
add("... in the reference's own axis order (axis_order = None)", spec.nS * 9 * out["stages_done"],
    out["sweep_ms"], v, "stopped after %d stages" % out["stages_done"])
# 6-D north-star figure: the attitude model on a 24^6 grid x 11^3 torques, tabulated next angles
import os
if os.environ.get("HJB_MEASURE_6D", "1") == "1":
    sa6 = hjbdp.Solver_attitude(n_mesh_w=24, n_mesh_q=24)
    sa6.U_vector = np.linspace(-0.11, 0.11, 11)
    t0 = time.time()
    spec6 = sa6.build_spec_full()
    pspec6, _ = hjbdp.permute_state_axes(spec6, sa6.AXIS_ORDER)
    t_host = time.time() - t0
    with hjbdp.Backup(pspec6) as bk:
        v = bk.info()["kernel_variant"]
        bk.solve(1)
        out = bk.solve(2)
    add("6-D attitude model 24^6 x 11^3, 2 stages, f32", pspec6.nS * pspec6.nU * 2, out["sweep_ms"], v,
        "host table build %.0f s" % t_host)
# C4: 120^4 x 9, terms cost mode
pa.cost_mode = "terms"
pa.n_mesh_x = pa.n_mesh_v = pa.n_mesh_t = pa.n_mesh_w = 120
sx, sv, st, sw = pa.grids()
spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, 6, 6, .5, .5, .1, pa.J2)
# as the mirror runs it by default: the library's axis labelling (axis_order "auto" -> hjb_problem_suggest_order)
assert pa.axis_order == "auto"
spec_ref = spec
spec, _ = pa._relabel(spec_ref)
out, v = solve_timed(spec, 20)
add("C4 pos-att 120^4 x 9, 20 stages, f32, the mirror's default axis order (x, theta, w, v)", spec.nS * 9 * 20, out["sweep_ms"], v,
    "%.3f ms per stage" % (out["sweep_ms"] / 20))
out, v = solve_timed(spec_ref, 5)
add("C4 in the reference's own axis order (x, v, theta, w) (axis_order = None), 5 stages", spec.nS * 9 * 5, out["sweep_ms"], v,
    "%.3f ms per stage" % (out["sweep_ms"] / 5))
spec16 = hjbdp.ProblemSpec(spec.knots, spec.m, spec.next_terms, spec.cost_terms, dtype=np.float32, index_base=1,
                           j_storage=np.float16)
out, v = solve_timed(spec16, 5)
add("C5 = C4 with float16 cost-to-go storage, 5 stages", spec.nS * 9 * 5, out["sweep_ms"], v)
print("| config | backups | sweep ms | backups/s | kernel variant | note |")
print("|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %.3g | %.2f | %.3g | %d | %s |" % r)
