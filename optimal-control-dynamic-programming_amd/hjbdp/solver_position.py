"""Solver_position - host mirror of position-control/Solver_position.m.

`simplified_run` (:94-150): three independent 2-D (x_i, v_i) sweeps with the three
thrust levels U_vector = [-0.26 0 0.26]; the `for k_s = N_stage-1:-1:1` loop
(:132-141) runs in libhjbdp.  The forward simulation `get_optimal_path`
(:189-311, RKF45 + orbital dynamics) is out of scope.

Reference quirk kept bit for bit (:157-186): RK4_x integrates xdynamics(v) = v but
feeds V + k*h/2 back as the "state", so x+ = x + h*(k1+2k2+2k3+k4)/6 with
k1 = v, k2 = v + k1*h/2, ... does NOT depend on u; RK4_v has all k equal to u/Mass.
All arithmetic is double, as in the reference.
"""
from __future__ import annotations

import math

import numpy as np

from .core import Backup, solve_many
from .matlab_compat import interp_nearest_point, sym_linspace_position
from .problem import ProblemSpec, Term


class NearestPolicy:
    """griddedInterpolant({s_x,s_v}, U_vector(U_idx), 'nearest') (:144-146)."""

    def __init__(self, knots, values):
        self.GridVectors = knots
        self.Values = values

    def __call__(self, *x):
        return interp_nearest_point(self.GridVectors, self.Values, x)

    def lookup_many(self, points, device=0):
        """Batched evaluation at points [nq, D] on the GPU (hjb_policy_lookup, 'nearest')."""
        from .core import policy_lookup
        return policy_lookup(self.GridVectors, self.Values, points, "nearest", device=device)


class Solver_position:
    def __init__(self):
        # Solver_position.m:46-92
        self.v_min, self.v_max = -0.5, 0.5
        self.x_min, self.x_max = -0.5, 0.5
        self.n_mesh_v = 200
        self.n_mesh_x = 200
        self.Mass = 4.16
        self.Qx1 = self.Qx2 = self.Qx3 = 6.0
        self.Qv1 = self.Qv2 = self.Qv3 = 6.0
        self.R1 = self.R2 = self.R3 = 0.1
        self.T_final = 30.0
        self.h = 0.005
        self._finish_init()
        self.defaultX0 = np.zeros(6)
        self.U_vector = np.array([-0.13, 0.0, 0.13]) * 2.0
        self.U1_Opt = self.U2_Opt = self.U3_Opt = None
        self.device = 0
        self.F_values = [None, None, None]   # F_i.Values after the sweep
        self.U_idx = [None, None, None]      # U_i_idx (1-based)
        self.sweep_ms = [None, None, None]

    def _finish_init(self):
        # :75-81 (isinteger() of a double is always false -> always ceil)
        self.N_stage = int(math.ceil(self.T_final / self.h))
        self.T_final = self.h * self.N_stage

    # -- Solver_position.m:152-186 on grid VECTORS (the 3-D arrays of the
    #    reference are ndgrid copies of these) ------------------------------
    def _dx_of_v(self, V, h):
        k1 = V
        k2 = V + k1 * h / 2
        k3 = V + k2 * h / 2
        k4 = V + k3 * h
        return h * (k1 + 2 * k2 + 2 * k3 + k4) / 6

    def _dv_of_u(self, U, h):
        k = U / self.Mass
        return h * (k + 2 * k + 2 * k + k) / 6

    def build_spec(self, channel):
        Qx = (self.Qx1, self.Qx2, self.Qx3)[channel]
        Qv = (self.Qv1, self.Qv2, self.Qv3)[channel]
        R = (self.R1, self.R2, self.R3)[channel]
        s_x = sym_linspace_position(self.x_min, self.x_max, self.n_mesh_x)   # :97-104
        s_v = sym_linspace_position(self.v_min, self.v_max, self.n_mesh_v)
        U = np.asarray(self.U_vector, dtype=np.float64)
        nxt = [[Term((0,), s_x), Term((1,), self._dx_of_v(s_v, self.h))],     # RK4_x :157-167
               [Term((1,), s_v), Term((2,), self._dv_of_u(U, self.h))]]       # RK4_v :173-182
        cost = [Term((0,), Qx * s_x ** 2), Term((1,), Qv * s_v ** 2), Term((2,), R * U ** 2)]  # :113
        return ProblemSpec([s_x, s_v], [len(U)], nxt, cost, dtype=np.float64, index_base=1), s_x, s_v

    def simplified_run(self, n_stages=None, keep_policy=False):
        """n_stages overrides N_stage-1 (tests).  keep_policy=True also leaves every stage's policy - the per-stage store of
        the reference's development scripts (attitude-control/test/test_simplified.m:102-104, `U1_Opt(:,:,k_s) =
        U_vector(U1_idx)`; test/Dynamic_Solver.m:100 keeps u_star_idxs the same way): self.U_Opt_stages[ch] is
        [n_x, n_v, n_stages] with stage k_s in plane k_s - 1, self.U_idx_stages[ch] the 1-based labels."""
        n_st = self.N_stage - 1 if n_stages is None else int(n_stages)
        built = [self.build_spec(ch) for ch in range(3)]
        # the three channels are independent sweeps (:132-141 runs them in one loop body): in flight together
        outs, self.wall_ms, _ = solve_many([b[0] for b in built], n_st, device=self.device, keep_idx=bool(keep_policy))
        self.U_Opt_stages = self.U_idx_stages = None
        if keep_policy:
            self.U_idx_stages = [outs[ch]["idx_stages"].reshape(len(built[ch][1]), len(built[ch][2]), n_st, order="F") for ch in range(3)]
            self.U_Opt_stages = [np.asarray(self.U_vector, dtype=np.float64)[ix - 1] for ix in self.U_idx_stages]
        for ch in range(3):
            spec, s_x, s_v = built[ch]
            out = outs[ch]
            shape = (len(s_x), len(s_v))
            self.F_values[ch] = out["J"].reshape(shape, order="F")
            self.U_idx[ch] = out["idx"].reshape(shape, order="F")
            self.sweep_ms[ch] = out["sweep_ms"]
            pol = NearestPolicy([s_x, s_v], self.U_vector[self.U_idx[ch] - 1])
            setattr(self, "U%d_Opt" % (ch + 1), pol)
        self.n_mesh_x, self.n_mesh_v = len(s_x), len(s_v)                     # :100,:104
        return self

    # ------------------------------------------------------------------ closed-loop rollout
    def get_target_R0V0(self):
        """Solver_position.m:313-331: the target's initial state on its reference orbit (perigee altitude 300 km,
        e = 0.1, equatorial, at perigee)."""
        from .orbit import MU_EARTH, R_EARTH, state_from_elements
        rp, e = R_EARTH + 300.0, 0.1
        ra = rp * (1.0 + e) / (1.0 - e)
        h_ = math.sqrt(2.0 * MU_EARTH * rp * ra / (ra + rp))
        return state_from_elements(h_, e, 0.0, 0.0, 0.0, 0.0, MU_EARTH)

    def get_optimal_path(self, n_steps=None, y0=None):
        """Solver_position.m:189-311 without the plots: from the chaser's initial relative state (default
        dr0 = [-1 0 0] km, dv0 = 0, :195-197) step N-1 times: look the three accelerations up in the
        'nearest' policies left by simplified_run (:215-217), hold them over [t_k, t_k + h] and integrate the
        relative-motion equations with RKF4(5) (:222).  Returns (T [N], X [6, N], F_Opt_history [3, N]);
        n_steps limits the number of steps (tests)."""
        from .orbit import relative_motion_rates, rkf45
        if self.U_idx[0] is None:
            raise RuntimeError("simplified_run() first")
        y = np.array([-1.0, 0.0, 0.0, 0.0, 0.0, 0.0]) if y0 is None else np.asarray(y0, dtype=np.float64).reshape(6)
        R0, V0 = self.get_target_R0V0()
        N = int(math.ceil(self.T_final / self.h))
        if n_steps is not None:
            N = min(N, int(n_steps) + 1)
        X = np.zeros((6, N))
        F = np.zeros((3, N))
        X[:, 0] = y
        for k in range(N - 1):
            xs = X[:, k]
            a = (float(self.U1_Opt(xs[0], xs[3])), float(self.U2_Opt(xs[1], xs[4])), float(self.U3_Opt(xs[2], xs[5])))
            F[:, k] = a
            X[:, k + 1] = rkf45(lambda t, yy: relative_motion_rates(t, yy, R0, V0, a), k * self.h, (k + 1) * self.h, xs)
        self.X_path, self.F_Opt_history = X, F
        return np.arange(N) * self.h, X, F

