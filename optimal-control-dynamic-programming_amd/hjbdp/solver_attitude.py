"""Solver_attitude - host mirror of attitude-control/Solver_attitude.m.

simplified_run (:196-259): three independent 2-D (w_i, theta_i) sweeps, double.
run (:261-300): the 6-D state (w1,w2,w3,yaw,pitch,roll) x 3-D torque problem in
single precision, operands reshaped onto dims 1..9 exactly like `reshape_states`
(:717-742).  The committed `run` cannot execute in MATLAB (it calls
calculate_J_U_opt_state_M with 2 arguments, :282 vs :384, and defaults to
n_mesh_w = 1000); its intended semantics are the specification here, with small
default sizes taken from the .asv revision.  The argmin returned is the true joint
minimiser (i1,i2,i3) in cascade order; the reference's index composition at
:290-292 is a linear-indexing bug and J is unaffected by it.
Forward simulations (:508-591, :670-696, :744-925) are out of scope.
"""
from __future__ import annotations

import math

import numpy as np

from .core import Backup, solve_batch, solve_many
from .matlab_compat import deg2rad, linspace
from .problem import ProblemSpec, Term
from .solver_position import NearestPolicy

f32 = np.float32


class Solver_attitude:
    def __init__(self, n_mesh_w=11, n_mesh_q=10, n_mesh_t=300, n_mesh_w_simplified=1000):
        # Solver_attitude.m:103-193 (n_mesh_w for run(): .asv size 11, see module doc)
        self.w_min = -float(deg2rad(50))
        self.w_max = -float(deg2rad(-50))
        self.n_mesh_w = int(n_mesh_w)
        self.n_mesh_w_simplified = int(n_mesh_w_simplified)   # committed default 1000 (:108) for simplified_run
        self.yaw_min, self.yaw_max = -30.0, 30.0
        self.pitch_min, self.pitch_max = -20.0, 20.0
        self.roll_min, self.roll_max = -35.0, 35.0
        self.n_mesh_q = int(n_mesh_q)
        self.n_mesh_t = int(n_mesh_t)
        i1, i2, i3 = 0.02836 + 0.00016, 0.026817 + 0.00150, 0.023 + 0.00150
        i4, i5, i6 = -0.0000837, 0.000014, -0.00029
        self.InertiaM = np.array([[i1, i4, i5], [i4, i2, i6], [i5, i6, i3]])
        self.Q1 = self.Q2 = self.Q3 = 6.0
        self.Q4 = self.Q5 = self.Q6 = 6.0
        self.R1 = self.R2 = self.R3 = 4.0
        self.Qt1, self.Qt2, self.Qt3 = self.Q4, self.Q5, self.Q6
        self.T_final = 30.0
        self.h = 0.005
        self.N_stage = int(math.ceil(self.T_final / self.h))
        self.T_final = self.h * self.N_stage
        self.J1, self.J2, self.J3 = self.InertiaM[0, 0], self.InertiaM[1, 1], self.InertiaM[2, 2]   # InertiaM(1),(5),(9)
        self.U_vector = np.array([-0.11, 0.0, 0.11])
        self.device = 0
        self.batch_channels = True     # simplified_run: the three channels as one launch per stage (False: three chains on threads)
        self.batch_groups = None
        self.U1_Opt = self.U2_Opt = self.U3_Opt = None
        self.F_values = None
        self.U_idx = None
        self.sweep_ms = None

    # ------------------------------------------------------------------ 2-D
    def _dw_of_u(self, U, J, h):          # RK4_w :630-644, wdynamics ignores w
        k = U / J
        return h * (k + 2 * k + 2 * k + k) / 6

    def _dt_of_w(self, W, h):             # RK4_t :646-660
        k1 = W
        k2 = W + k1 * h / 2
        k3 = W + k2 * h / 2
        k4 = W + k3 * h
        return h * (k1 + 2 * k2 + 2 * k3 + k4) / 6

    def build_spec_simplified(self, channel):
        s_w = linspace(self.w_min, self.w_max, self.n_mesh_w_simplified)                       # :199-201
        lims = [(self.yaw_min, self.yaw_max), (self.pitch_min, self.pitch_max), (self.roll_min, self.roll_max)][channel]
        s_t = linspace(float(deg2rad(lims[0])), float(deg2rad(lims[1])), self.n_mesh_t)         # :203-205
        Jc = (self.J1, self.J2, self.J3)[channel]
        Qw = (self.Q1, self.Q2, self.Q3)[channel]
        Qt = (self.Qt1, self.Qt2, self.Qt3)[channel]
        R = (self.R1, self.R2, self.R3)[channel]
        U = self.U_vector
        nxt = [[Term((0,), s_w), Term((2,), self._dw_of_u(U, Jc, self.h))],                     # w_next
               [Term((1,), s_t), Term((0,), self._dt_of_w(s_w, self.h))]]                       # t_next = T + f(W)
        cost = [Term((0,), Qw * s_w ** 2), Term((1,), Qt * s_t ** 2), Term((2,), R * U ** 2)]   # :220
        return ProblemSpec([s_w, s_t], [len(U)], nxt, cost, dtype=np.float64, index_base=1), s_w, s_t

    def simplified_run(self, n_stages=None, keep_policy=False):
        """keep_policy=True also leaves the policy of EVERY stage, as the reference's development script stores it
        (attitude-control/test/test_simplified.m:102-104: `U1_Opt(:,:,k_s) = U_vector(U1_idx)`): self.U_Opt_stages[ch] is
        the [n_w, n_t, n_stages] array of torque values with stage k_s in plane k_s - 1 (and self.U_idx_stages[ch] the
        1-based labels), written by the stage kernel itself (hjb_solve_opts.idx_stages)."""
        n_st = self.N_stage - 1 if n_stages is None else int(n_stages)
        self.F_values, self.U_idx, self.sweep_ms = [None] * 3, [None] * 3, [None] * 3
        self.U_Opt_stages = self.U_idx_stages = None
        built = [self.build_spec_simplified(ch) for ch in range(3)]
        if keep_policy or not self.batch_channels:
            outs, self.wall_ms, _ = solve_many([b[0] for b in built], n_st, device=self.device,   # channels side by side
                                               keep_idx=bool(keep_policy))
            self.batch_groups = [1, 1, 1]
        else:                              # ... as ONE launch per stage for the three (hjb_solve_batch)
            outs, self.wall_ms, _, self.batch_groups = solve_batch([b[0] for b in built], n_st, device=self.device)
        if keep_policy:
            self.U_idx_stages = [outs[ch]["idx_stages"].reshape(len(built[ch][1]), len(built[ch][2]), n_st, order="F") for ch in range(3)]
            self.U_Opt_stages = [self.U_vector[ix - 1] for ix in self.U_idx_stages]
        for ch in range(3):
            spec, s_w, s_t = built[ch]
            out = outs[ch]
            shape = (len(s_w), len(s_t))
            self.F_values[ch] = out["J"].reshape(shape, order="F")
            self.U_idx[ch] = out["idx"].reshape(shape, order="F")
            self.sweep_ms[ch] = out["sweep_ms"]
            setattr(self, "U%d_Opt" % (ch + 1), NearestPolicy([s_w, s_t], self.U_vector[self.U_idx[ch] - 1]))  # :249-251
        return self

    # ------------------------------------------------------------------ 6-D
    def build_spec_full(self):
        """Single-precision operands with MATLAB's typing (typed properties :83-96:
        X*V, cang/sang, U*V are single; scalars are double -> single*double = single)."""
        nw, nq = self.n_mesh_w, self.n_mesh_q
        sr = [linspace(self.w_min, self.w_max, nw)] * 3                                          # :177-179
        s_yaw = linspace(float(deg2rad(self.yaw_min)), float(deg2rad(self.yaw_max)), nq)        # :181-183
        s_pitch = linspace(float(deg2rad(self.pitch_min)), float(deg2rad(self.pitch_max)), nq)
        s_roll = linspace(float(deg2rad(self.roll_min)), float(deg2rad(self.roll_max)), nq)
        # reshape_states :717-742 (single typed properties)
        X1V, X2V, X3V = (s.astype(f32) for s in sr)
        c4, s4 = np.cos(s_yaw / 2).astype(f32), np.sin(s_yaw / 2).astype(f32)
        c5, s5 = np.cos(s_pitch / 2).astype(f32), np.sin(s_pitch / 2).astype(f32)
        c6, s6 = np.cos(s_roll / 2).astype(f32), np.sin(s_roll / 2).astype(f32)
        UV = self.U_vector.astype(f32)
        h, J1, J2, J3 = f32(self.h), self.J1, self.J2, self.J3
        # broadcast helpers over (yaw, pitch, roll) = dims 3,4,5
        C4, S4 = c4[:, None, None], s4[:, None, None]
        C5, S5 = c5[None, :, None], s5[None, :, None]
        C6, S6 = c6[None, None, :], s6[None, None, :]
        q1 = S4 * C5 * C6 - C4 * S5 * S6         # x4 (:318,:419)
        q2 = C4 * S5 * C6 + S4 * C5 * S6         # x5
        q3 = C4 * C5 * S6 - S4 * S5 * C6         # x6
        x7 = (f32(1) - (q1 ** 2 + q2 ** 2 + q3 ** 2)) ** f32(0.5)            # :419-421
        # w_next :423-425   X1V + h*((J2-J3)/J1*X2V.*X3V + U1V/J1)
        def wnext(c, Xb, Xc, JJ):
            inner = f32(c) * Xb[:, None, None] * Xc[None, :, None] + (UV / f32(JJ))[None, None, :]
            return (h * inner).astype(f32)
        t1 = wnext((J2 - J3) / J1, X2V, X3V, J1)      # dims (1,2,6)
        t2 = wnext((J3 - J1) / J2, X3V, X1V, J2)      # built as (X3,X1,U2) -> dims (2,0,7): reorder to (0,2,7)
        t3 = wnext((J1 - J2) / J3, X1V, X2V, J3)      # dims (0,1,8)
        t2 = np.ascontiguousarray(np.transpose(t2, (1, 0, 2)))
        # quaternion kinematics + renormalise + back to Euler angles :449-493 (single)
        W1 = X1V[:, None, None, None, None, None]
        W2 = X2V[None, :, None, None, None, None]
        W3 = X3V[None, None, :, None, None, None]
        Q1, Q2, Q3, Q7 = (a[None, None, None] for a in (q1, q2, q3, x7))
        half = f32(0.5)
        X4n = Q1 + h * (half * (W3 * Q2 - W2 * Q3 + W1 * Q7))                 # :449-452
        X5n = Q2 + h * (half * (-W3 * Q1 + W1 * Q3 + W2 * Q7))                # :454-457
        X6n = Q3 + h * (half * (W2 * Q1 - W1 * Q2 + W3 * Q7))                 # :459-462
        X7n = Q7 + h * (half * (-W1 * Q1 - W2 * Q2 - W3 * Q3))                # :465-467
        nrm = np.sqrt(X4n ** 2 + X5n ** 2 + X6n ** 2 + X7n ** 2)              # :477
        X4n, X5n, X6n, X7n = X4n / nrm, X5n / nrm, X6n / nrm, X7n / nrm
        two = f32(2)
        yaw_n = np.arctan2(two * (X6n * X5n + X7n * X4n), X7n ** 2 + X6n ** 2 - X5n ** 2 - X4n ** 2)      # :485-486
        pitch_n = np.arcsin(-two * (X6n * X4n - X7n * X5n))                                               # :487
        roll_n = np.arctan2(two * (X5n * X4n + X7n * X6n), X7n ** 2 - X6n ** 2 - X5n ** 2 + X4n ** 2)     # :488-489
        st6 = (0, 1, 2, 3, 4, 5)
        nxt = [[Term((0,), X1V), Term((1, 2, 6), t1)],
               [Term((1,), X2V), Term((0, 2, 7), t2)],
               [Term((2,), X3V), Term((0, 1, 8), t3)],
               [Term(st6, yaw_n.astype(f32))], [Term(st6, pitch_n.astype(f32))], [Term(st6, roll_n.astype(f32))]]
        # cost :316-321
        cost = [Term((0,), f32(self.Q1) * X1V ** 2), Term((1,), f32(self.Q2) * X2V ** 2), Term((2,), f32(self.Q3) * X3V ** 2),
                Term((3, 4, 5), f32(self.Q4) * q1 ** 2), Term((3, 4, 5), f32(self.Q5) * q2 ** 2),
                Term((3, 4, 5), f32(self.Q6) * q3 ** 2),
                Term((6,), f32(self.R1) * UV ** 2), Term((7,), f32(self.R2) * UV ** 2), Term((8,), f32(self.R3) * UV ** 2)]
        knots = [sr[0], sr[1], sr[2], s_yaw, s_pitch, s_roll]
        knots = [k.astype(f32).astype(np.float64) for k in knots]      # F grid vectors with single Values: single arithmetic
        nu = len(UV)
        return ProblemSpec(knots, [nu, nu, nu], nxt, cost, dtype=np.float32, index_base=1)

    def build_spec_model(self, j_storage=None):
        """The same problem as `permute_state_axes(build_spec_full(), AXIS_ORDER)` - axes (yaw, pitch, roll, w1, w2,
        w3) - but WITHOUT the three nS-sized next-angle tables: the quaternion kinematics :449-489 are evaluated
        inside the library per state (hjbdp.h HJB_MODEL_QUAT_EULER321) from the four nq^3 quaternion tables.
        This is what makes 51^6 possible: every table below is O(n^3).  The next angles
        then come from the library's fixed polynomial atan2/asin instead of numpy's libm, so results agree with
        the table form to rounding (a few ulp in x_next), not bit for bit."""
        nw, nq = self.n_mesh_w, self.n_mesh_q
        sr = [linspace(self.w_min, self.w_max, nw)] * 3
        s_yaw = linspace(float(deg2rad(self.yaw_min)), float(deg2rad(self.yaw_max)), nq)
        s_pitch = linspace(float(deg2rad(self.pitch_min)), float(deg2rad(self.pitch_max)), nq)
        s_roll = linspace(float(deg2rad(self.roll_min)), float(deg2rad(self.roll_max)), nq)
        X1V, X2V, X3V = (s.astype(f32) for s in sr)
        c4, s4 = np.cos(s_yaw / 2).astype(f32), np.sin(s_yaw / 2).astype(f32)
        c5, s5 = np.cos(s_pitch / 2).astype(f32), np.sin(s_pitch / 2).astype(f32)
        c6, s6 = np.cos(s_roll / 2).astype(f32), np.sin(s_roll / 2).astype(f32)
        UV = self.U_vector.astype(f32)
        h, J1, J2, J3 = f32(self.h), self.J1, self.J2, self.J3
        C4, S4 = c4[:, None, None], s4[:, None, None]
        C5, S5 = c5[None, :, None], s5[None, :, None]
        C6, S6 = c6[None, None, :], s6[None, None, :]
        q1 = S4 * C5 * C6 - C4 * S5 * S6
        q2 = C4 * S5 * C6 + S4 * C5 * S6
        q3 = C4 * C5 * S6 - S4 * S5 * C6
        x7 = (f32(1) - (q1 ** 2 + q2 ** 2 + q3 ** 2)) ** f32(0.5)

        def wnext(c, Xb, Xc, JJ):
            inner = f32(c) * Xb[:, None, None] * Xc[None, :, None] + (UV / f32(JJ))[None, None, :]
            return (h * inner).astype(f32)
        t1 = wnext((J2 - J3) / J1, X2V, X3V, J1)                                        # (w2, w3, U1) = dims (4,5,6)
        t2 = np.ascontiguousarray(np.transpose(wnext((J3 - J1) / J2, X3V, X1V, J2), (1, 0, 2)))   # (w1, w3, U2) = (3,5,7)
        t3 = wnext((J1 - J2) / J3, X1V, X2V, J3)                                        # (w1, w2, U3) = (3,4,8)
        nxt = [[], [], [],
               [Term((3,), X1V), Term((4, 5, 6), t1)],
               [Term((4,), X2V), Term((3, 5, 7), t2)],
               [Term((5,), X3V), Term((3, 4, 8), t3)]]
        cost = [Term((3,), f32(self.Q1) * X1V ** 2), Term((4,), f32(self.Q2) * X2V ** 2), Term((5,), f32(self.Q3) * X3V ** 2),
                Term((0, 1, 2), f32(self.Q4) * q1 ** 2), Term((0, 1, 2), f32(self.Q5) * q2 ** 2),
                Term((0, 1, 2), f32(self.Q6) * q3 ** 2),
                Term((6,), f32(self.R1) * UV ** 2), Term((7,), f32(self.R2) * UV ** 2), Term((8,), f32(self.R3) * UV ** 2)]
        knots = [s_yaw, s_pitch, s_roll, sr[0], sr[1], sr[2]]
        knots = [k.astype(f32).astype(np.float64) for k in knots]
        nu = len(UV)
        return ProblemSpec(knots, [nu, nu, nu], nxt, cost, dtype=np.float32, index_base=1, j_storage=j_storage,
                           model={"kind": "quat_euler321", "h": float(h), "tables": [q1, q2, q3, x7]})

    # state-axis labelling handed to the library: the three angle axes (whose next value does not depend on
    # the torques) FIRST, then w1, w2, w3 (driven by U1 outermost ... U3 innermost, w3 LAST): the stage kernel
    # then contracts the angle axes once per state (variant 4, mode 2) instead of once per torque pair.
    # Results are mapped back to the reference's dim order.
    AXIS_ORDER = (3, 4, 5, 0, 1, 2)

    def run(self, n_stages=None, relabel=True, on_the_fly=False):
        """on_the_fly=True: the next angles are computed inside the library (build_spec_model) instead of being
        tabulated over the whole 6-D grid - the form that scales to 51^6."""
        n_st = self.N_stage - 1 if n_stages is None else int(n_stages)
        if on_the_fly:
            pspec = self.build_spec_model()
            shape = tuple(pspec.n[a] for a in (3, 4, 5, 0, 1, 2))

            def to_old(flat):
                return np.transpose(np.asarray(flat).reshape(pspec.n, order="F"), (3, 4, 5, 0, 1, 2)).reshape(-1, order="F")
        else:
            spec = self.build_spec_full()
            shape = spec.n
            if relabel:
                from .problem import permute_state_axes
                pspec, to_old = permute_state_axes(spec, self.AXIS_ORDER)
            else:
                pspec, to_old = spec, (lambda x: x)
        with Backup(pspec, device=self.device) as bk:
            out = bk.solve(n_st)
            self.kernel_variant = bk.info()["kernel_variant"]
        self.F_values = to_old(out["J"]).reshape(shape, order="F")
        lab = to_old(out["idx"]).reshape(shape, order="F") - 1
        nu = len(self.U_vector)
        i1, i2, i3 = lab % nu, (lab // nu) % nu, lab // (nu * nu)
        self.U_idx = (i1 + 1, i2 + 1, i3 + 1)
        self.sweep_ms = out["sweep_ms"]
        # :296-298  U_i_Opt = single(U_vector(U_i_Opt))
        self.U1_Opt = self.U_vector[i1].astype(f32)
        self.U2_Opt = self.U_vector[i2].astype(f32)
        self.U3_Opt = self.U_vector[i3].astype(f32)
        return self

    # ------------------------------------------------------------------ closed-loop rollouts (hjbdp/rollout.py)
    def grid_vectors_full(self):
        """The six grid vectors of `run` in the reference's dim order (w1, w2, w3, yaw, pitch, roll), :177-183."""
        nw, nq = self.n_mesh_w, self.n_mesh_q
        sr = linspace(self.w_min, self.w_max, nw)
        return [sr, sr, sr,
                linspace(float(deg2rad(self.yaw_min)), float(deg2rad(self.yaw_max)), nq),
                linspace(float(deg2rad(self.pitch_min)), float(deg2rad(self.pitch_max)), nq),
                linspace(float(deg2rad(self.roll_min)), float(deg2rad(self.roll_max)), nq)]

    def spacecraft_dynamics_list(self, X, U):
        from . import rollout
        return rollout.spacecraft_dynamics_list(self, X, U)                   # :600-620

    def next_stage_states(self, X1, U, h, mode="RK4"):
        from . import rollout
        return rollout.next_stage_states(self, X1, U, h, mode)                # :670-696

    def linear_control_response(self, X0=None, T_final=None, dt=None):
        from . import rollout
        return rollout.linear_control_response(self, X0, T_final, dt)         # :508-591

    def get_optimal_path(self, X0=None, method="nearest", n_steps=None):
        from . import rollout
        return rollout.attitude_optimal_path(self, X0, method, n_steps)       # :744-833 (after run)

    def get_optimal_path_simplified_testode45(self, X0=None, n_steps=None):
        from . import rollout
        return rollout.attitude_optimal_path_simplified(self, X0, n_steps)    # :835-925 (after simplified_run)

