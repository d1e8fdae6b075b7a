"""Solver_pos_att - host mirror of pos-att/Solver_pos_att.m (the sweep part).

simplified_run (:197-242) -> four calls of calculate_one_channel_U_Opt (:244-297):
per channel a 4-D state (x, v, theta, w) x 9 thruster combinations (6 for the
thruster-failure channel), `sym_linspace` grids (:906-918, non-uniform for even n),
first-order ("RK4_*" are Euler, :330-402) next states, cost of J_current_reshaped
(:784-802), 1999 stages with the early-stop monitor every 50 stages (:268-285,
restated as intended: idsum50_prev is read before assignment in the reference).
The reference saves each controller to a .mat file (:289); here the same four
variables are kept in `self.controllers[file_name]`.

Typing: the reference keeps J single (`zeros(...,'single')`, :265) with DOUBLE
grid vectors and query tables (x_next .. w_next, :299-327).  `table_dtype =
np.float64` (the default here) is that typing: the next-state terms go to the
library in float64, each query is formed, located and weighted in double and the
weight rounded to float32 once (hjbdp.h HJB_TAB_F64) - the (cell, weight) tables
are stage-invariant, so it costs nothing per stage; `table_dtype = None` runs
float32 end to end.  The stage cost: cost_mode='f64' (default) passes the five
separable operands in double, summed in double and rounded to single once per
(state, control) - bit-identical to `J_current_M = single(...)` (:800-801) without
the nS*nU array; cost_mode='exact' passes that array itself as ONE full-mask term;
cost_mode='terms' the five operands in float32 (float32 sums, ~1 ulp different).
Axis order: by default the library's suggestion (x, theta, w, v) with results
mapped back; axis_order=None runs the reference's own (x, v, theta, w).
`monitor_single` (default True): the early-stop monitor's `sum(F_gI.Values(:))`
of a single array is a single-precision sum in MATLAB (:274); its order there is
not documented, the library's is (csrc/kernels_reduce.h).  U_Optimal_id is kept
in the narrowest label type (uint8: nine thruster combinations).
Policy use / forward simulation (:404-847): hjbdp/rollout.py (host-side, get_optimal_path below).
"""
from __future__ import annotations

import math

import numpy as np

from .core import Backup, solve_batch, solve_many
from .matlab_compat import deg2rad, sym_linspace_pos_att
from .problem import ProblemSpec, Term


def vectors_allcomb(f1, f2, f3, f4):
    """Solver_pos_att.m:886-904: ndgrid of the four thruster level vectors in
    column-major order, minus combinations firing opposing thrusters
    (f1>0 & f3<0) or (f2>0 & f4<0)."""
    f1, f2, f3, f4 = (np.atleast_1d(np.asarray(f, dtype=np.float64)) for f in (f1, f2, f3, f4))
    G = np.meshgrid(f1, f2, f3, f4, indexing="ij")
    F1, F2, F3, F4 = (g.reshape(-1, order="F") for g in G)
    rm = ((F1 > 0) & (F3 < 0)) | ((F2 > 0) & (F4 < 0))
    keep = ~rm
    return F1[keep], F2[keep], F3[keep], F4[keep]


class Solver_pos_att:
    # Relabelling of the state axes (x, v, theta, w) -> (x, theta, w, v) for large grids: the two axes whose next value
    # does not depend on the thrusters (x+ over (x, v), theta+ over (theta, w), :299-328) come first, which is the
    # shape of libhjbdp's column-sweep stage kernel (csrc/kernels_colsweep.h); of the two thruster-driven axes the
    # less-moved one (v: < 1 cell per stage; w moves up to 8 on a 120^4 grid) comes last, where a multi-GPU run shards
    # with a one-plane halo.  It is what hjb_problem_suggest_order returns for a channel (tests/test_host_solvers.py),
    # and what axis_order = "auto" asks the library for.  Pure bookkeeping: the 1-D lerps of the interpolation are
    # taken in the new order (results agree with the reference order to a few ulp per stage).
    FAST_AXIS_ORDER = (0, 2, 3, 1)

    def __init__(self):
        # Solver_pos_att.m:96-195
        self.v_min, self.v_max, self.n_mesh_v = -0.1, 0.1, 30
        self.x_min, self.x_max, self.n_mesh_x = -0.2, 0.2, 30
        self.w_min, self.w_max, self.n_mesh_w = float(deg2rad(-2)), float(deg2rad(2)), 15
        self.theta1_min, self.theta1_max = -5.0, 5.0
        self.theta2_min, self.theta2_max = -6.0, 6.0
        self.theta3_min, self.theta3_max = -7.0, 7.0
        self.n_mesh_t = 20
        self.Mass = 4.16
        i1, i2, i3 = 0.02836 + 0.00016, 0.026817 + 0.00150, 0.023 + 0.00150
        i4, i5, i6 = -0.0000837, 0.000014, -0.00029
        self.InertiaM = np.array([[i1, i4, i5], [i4, i2, i6], [i5, i6, i3]])
        self.J1, self.J2, self.J3 = self.InertiaM[0, 0], self.InertiaM[1, 1], self.InertiaM[2, 2]
        self.Qx1 = self.Qx2 = self.Qx3 = 6.0
        self.Qv1 = self.Qv2 = self.Qv3 = 6.0
        self.Qt1 = self.Qt2 = self.Qt3 = 0.5
        self.Qw1 = self.Qw2 = self.Qw3 = 0.5
        self.R1 = self.R2 = self.R3 = 0.1
        self.T_final = 10.0
        self.h = 0.005
        self.N_stage = int(math.ceil(self.T_final / self.h))
        self.T_final = self.h * self.N_stage
        T = 0.13
        self.T_dist = 9.65e-2
        self.F_Thr0 = self.F_Thr1 = self.F_Thr2 = self.F_Thr3 = self.F_Thr4 = self.F_Thr5 = np.array([0.0, T])
        self.F_Thr6 = self.F_Thr7 = self.F_Thr8 = self.F_Thr9 = self.F_Thr10 = self.F_Thr11 = -np.array([0.0, T])
        self.monitor_period = 50      # :273
        self.monitor_tol = 1e-2       # :269
        # Defaults = the FAST path: the library's own axis labelling and the separable double cost operands.
        # cost_mode 'f64' is bit-identical to the reference's materialised single(double sum) ('exact') without the nS x nU
        # array; axis_order "auto" asks hjb_problem_suggest_order - (x, theta, w, v) for a channel: the column-sweep kernel,
        # 2 - 4x faster than the kernels the reference order runs on - and maps J / U_Optimal_id back to (x, v, theta, w).
        # The reference's own order is the opt-in: axis_order = None (J then differs from the default's by the order of the
        # 1-D lerps: <= 4e-5 of max J after the full sweep, 99.98 % equal labels; neither order is pinned by a MATLAB artefact
        # beyond 2-D).  Both are held to the oracle bit for bit (tests/test_gpu_solvers.py).
        self.cost_mode = "f64"        # 'exact' | 'terms' | 'f64'
        self.axis_order = "auto"      # "auto" | FAST_AXIS_ORDER | None (the reference's order)
        self.table_dtype = np.float64 # the reference's typing of the query tables (:299-327); None = float32 queries
        self.idx_dtype = "auto"       # U_Optimal_id storage: uint8 for the 9 (6) thruster combinations
        self.monitor_single = True    # sum(F_gI.Values(:)) as a single-precision sum (:274)
        self.device = 0
        self.batch_channels = True    # simplified_run: the four channels as one launch per stage where the library can (hjb_solve_batch)
        self.batched = False          # ... whether the last simplified_run did (batch_groups: how many channels each launch chain carried)
        self.batch_groups = None
        self.controllers = {}

    # ------------------------------------------------------------------
    def grids(self):
        sx = sym_linspace_pos_att(self.x_min, self.x_max, self.n_mesh_x)
        sv = sym_linspace_pos_att(self.v_min, self.v_max, self.n_mesh_v)
        st = [sym_linspace_pos_att(float(deg2rad(a)), float(deg2rad(b)), self.n_mesh_t)
              for a, b in ((self.theta1_min, self.theta1_max), (self.theta2_min, self.theta2_max),
                           (self.theta3_min, self.theta3_max))]
        sw = sym_linspace_pos_att(self.w_min, self.w_max, self.n_mesh_w)
        return sx, sv, st, sw

    def build_channel_spec(self, s_x, s_v, s_t, s_w, f0, f1, f6, f7, Qx, Qv, Qt, Qw, R, J):
        """One calculate_one_channel_U_Opt problem (:244-265)."""
        fa, fb, fc, fd = vectors_allcomb(f0, f1, f6, f7)                  # :253
        h, d = self.h, self.T_dist
        # next_stage_states_simplified :299-328 with the Euler steps :330-402 (double)
        dv = h * ((fa + fb + fc + fd) / self.Mass)
        dw = h * ((fa * d + fb * (-d) + fc * d + fd * (-d)) / J)
        f32 = np.float32
        nxt = [[Term((0,), s_x), Term((1,), h * s_v)],
               [Term((1,), s_v), Term((4,), dv)],
               [Term((2,), s_t), Term((3,), h * s_w)],
               [Term((3,), s_w), Term((4,), dw)]]
        # J_current_reshaped :784-802 (argument order x,v,t,w; sum order Qx,Qv,Qw,Qt,R)
        cu = R * fa ** 2 + R * fb ** 2 + R * fc ** 2 + R * fd ** 2
        if self.cost_mode == "exact":
            X = s_x[:, None, None, None, None]
            V = s_v[None, :, None, None, None]
            Tt = s_t[None, None, :, None, None]
            W = s_w[None, None, None, :, None]
            full = (Qx * X ** 2 + Qv * V ** 2 + Qw * W ** 2 + Qt * Tt ** 2 + cu[None, None, None, None, :]).astype(f32)
            cost = [Term((0, 1, 2, 3, 4), full)]
        elif self.cost_mode in ("terms", "f64"):
            # 'terms': the five separable operands of :800 summed in single inside the library (<= 2 ulp from the double sum);
            # 'f64': the same operands kept in double, summed in double, ONE rounding per (state, control) - the reference's
            # single(double expression), bit-identical to 'exact' without the nS x nU array (hjbdp.h HJB_COST_F64)
            cost = [Term((0,), Qx * s_x ** 2), Term((1,), Qv * s_v ** 2), Term((3,), Qw * s_w ** 2),
                    Term((2,), Qt * s_t ** 2), Term((4,), cu)]
        else:
            raise ValueError("cost_mode must be 'exact', 'terms' or 'f64'")
        spec = ProblemSpec([s_x, s_v, s_t, s_w], [len(fa)], nxt, cost, dtype=np.float32, index_base=1,
                           idx_dtype=self.idx_dtype, table_dtype=self.table_dtype,
                           cost_dtype=np.float64 if self.cost_mode == "f64" else None)
        return spec, (fa, fb, fc, fd)

    def calculate_one_channel_U_Opt(self, s_x, s_v, s_t, s_w, f0, f1, f6, f7, Qx, Qv, Qt, Qw, R, J, file_name,
                                    n_stages=None, progress=None):
        spec, combos = self.build_channel_spec(s_x, s_v, s_t, s_w, f0, f1, f6, f7, Qx, Qv, Qt, Qw, R, J)
        n_st = self.N_stage - 1 if n_stages is None else int(n_stages)
        run_spec, to_old = self._relabel(spec)
        with Backup(run_spec, device=self.device) as bk:
            out = bk.solve(n_st, monitor_period=self.monitor_period, monitor_tol=self.monitor_tol, progress=progress,
                           monitor_single=self.monitor_single)
        out = self._map_back(out, to_old)
        return self._store_controller(file_name, (s_x, s_v, s_t, s_w), spec.n, combos, out)

    def _relabel(self, spec):
        order = self.axis_order
        if order == "auto":
            from .core import suggest_axis_order
            order = suggest_axis_order(spec)
        if order is None:
            return spec, None
        from .problem import permute_state_axes
        return permute_state_axes(spec, order)

    @staticmethod
    def _map_back(out, to_old):
        if to_old is not None:
            out = dict(out)
            out["J"], out["idx"] = to_old(out["J"]), to_old(out["idx"])
        return out

    def _store_controller(self, file_name, grids, shape, combos, out):
        self.controllers[file_name] = {                                  # save(file_name, ...) :289
            "GridVectors": list(grids),
            "F_gI_Values": out["J"].reshape(shape, order="F"),
            "U_Optimal_id": out["idx"].reshape(shape, order="F"),
            "f0_allcomb": combos[0], "f1_allcomb": combos[1], "f6_allcomb": combos[2], "f7_allcomb": combos[3],
            "stages_done": out["stages_done"], "stopped_early": out["stopped_early"], "sweep_ms": out["sweep_ms"],
        }
        return self.controllers[file_name]

    # ---- controller artefacts (Solver_pos_att.m:289 save, :849-884 set_controller) ----------
    def save_controllers(self, directory):
        """Write one MATLAB-loadable v5 .mat per channel with the variables the reference saves
        (`U_Optimal_id`, `f0_allcomb`, `f1_allcomb`, `f6_allcomb`, `f7_allcomb`); the interpolant
        object `F_gI` cannot be serialised outside MATLAB, so its two properties are stored as
        `F_gI_GridVectors` (1x4 cell) and `F_gI_Values` (single)."""
        import os
        import scipy.io
        os.makedirs(directory, exist_ok=True)
        paths = []
        for name, c in self.controllers.items():
            gv = np.empty((1, 4), dtype=object)
            for i, g in enumerate(c["GridVectors"]):
                gv[0, i] = np.asarray(g, dtype=np.float64).reshape(1, -1)
            path = os.path.join(directory, name + ".mat")
            scipy.io.savemat(path, {
                "F_gI_GridVectors": gv, "F_gI_Values": c["F_gI_Values"].astype(np.float32),
                "U_Optimal_id": c["U_Optimal_id"].astype(np.float64),     # MATLAB's min returns double indices
                "f0_allcomb": c["f0_allcomb"].reshape(-1, 1), "f1_allcomb": c["f1_allcomb"].reshape(-1, 1),
                "f6_allcomb": c["f6_allcomb"].reshape(-1, 1), "f7_allcomb": c["f7_allcomb"].reshape(-1, 1)})
            paths.append(path)
        return paths

    def set_controller(self, file, channel):
        """Solver_pos_att.m:849-884: four 'nearest' interpolants of the thruster levels selected by
        U_Optimal_id, stored on the object under the channel's thruster names."""
        import scipy.io
        from .solver_position import NearestPolicy
        Cc = scipy.io.loadmat(file)
        gv = [np.asarray(g, dtype=np.float64).reshape(-1) for g in Cc["F_gI_GridVectors"][0]]
        idx = Cc["U_Optimal_id"].astype(np.int64) - 1
        pols = [NearestPolicy(gv, Cc[k].reshape(-1)[idx]) for k in ("f0_allcomb", "f1_allcomb", "f6_allcomb", "f7_allcomb")]
        names = {"x": (0, 1, 6, 7), "y": (2, 3, 8, 9), "z": (4, 5, 10, 11)}
        if channel not in names:
            raise ValueError("wrong channel, must be one of x-y-z values")
        for pol, t in zip(pols, names[channel]):
            setattr(self, "Opt_F_Thr%d" % t, pol)
        return pols

    def simplified_run(self, n_stages=None, progress=None):
        """Solver_pos_att.m:197-242: the x, y, z channels and the thruster-failure variant of the x channel.  The
        four sweeps are independent, so they are in flight together (hjbdp.core.solve_many); results are stored
        per channel exactly as calculate_one_channel_U_Opt stores them."""
        sx, sv, st, sw = self.grids()
        jobs = [
            ((sx, sv, st[0], sw, self.F_Thr0, self.F_Thr1, self.F_Thr6, self.F_Thr7,
              self.Qx1, self.Qv1, self.Qt1, self.Qw1, self.R1, self.J2), "channel_x_controller_1"),          # :217-221
            ((sx, sv, st[1], sw, self.F_Thr2, self.F_Thr3, self.F_Thr8, self.F_Thr9,
              self.Qx2, self.Qv2, self.Qt2, self.Qw2, self.R2, self.J3), "channel_y_controller_1"),          # :223-227
            ((sx, sv, st[2], sw, self.F_Thr4, self.F_Thr5, self.F_Thr10, self.F_Thr11,
              self.Qx3, self.Qv3, self.Qt3, self.Qw3, self.R3, self.J1), "channel_z_controller_1"),          # :229-233
            ((sx, sv, st[0], sw, [0.0], self.F_Thr1, self.F_Thr6, self.F_Thr7,
              self.Qx1, self.Qv1, self.Qt1, self.Qw1, self.R1, self.J2), "channel_x_controller_1_failure"),  # :236-240
        ]
        built = [self.build_channel_spec(*args) for args, _ in jobs]
        n_st = self.N_stage - 1 if n_stages is None else int(n_stages)
        rel = [self._relabel(b[0]) for b in built]
        kw = dict(device=self.device, monitor_period=self.monitor_period, monitor_tol=self.monitor_tol, progress=progress,
                  monitor_single=self.monitor_single)
        self.batched, self.batch_groups = False, None
        if self.batch_channels:
            # channels of one column-sweep shape as ONE launch per stage (hjb_solve_batch), the groups side by side
            outs, self.wall_ms, _, self.batch_groups = solve_batch([r[0] for r in rel], n_st, **kw)
            self.batched = max(self.batch_groups) > 1
        else:
            outs, self.wall_ms, _ = solve_many([r[0] for r in rel], n_st, **kw)
        for (args, name), (spec, combos), out, r in zip(jobs, built, outs, rel):
            self._store_controller(name, args[:4], spec.n, combos, self._map_back(out, r[1]))
        return self

    # ------------------------------------------------------------------ closed-loop rollout (hjbdp/rollout.py)
    def get_target_R0V0(self):
        from . import rollout
        return rollout.target_R0V0()                                          # :734-753

    def to_Moments_Forces(self, f, R0, V0, q):
        from . import rollout
        return rollout.to_Moments_Forces(self, f, R0, V0, q)                  # :804-823

    def get_thruster_on_off_optimal(self, x, v, t, w, R0, V0, q):
        from . import rollout
        return rollout.get_thruster_on_off_optimal(rollout.thruster_policies(self), x, v, t, w, R0, V0, q)   # :404-449

    def get_optimal_path(self, X0=None, n_steps=None):
        """:452-730 without the plots -> (T, X [N, 13], F_Th_Opt [N, 12], Force_Moment_log [N, 6])."""
        from . import rollout
        return rollout.pos_att_optimal_path(self, X0, n_steps)

