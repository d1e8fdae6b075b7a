"""Dynamic_Solver - host mirror of the reference's Kirk Ch.3 two-state example
(test/Dynamic_Solver.m).  Same property names, same methods:

    objA = Dynamic_Solver(); objA.run(); objA.get_optimal_path()

`run` (Dynamic_Solver.m:66-105) builds the grid vectors and hands the backward
sweep - the `for k=1:N-1` loop around J_state_M (:202-220) - to libhjbdp; the
dx*dx*du tables of a_D_M (:184-188) and g_D (:196-200) are never materialised,
they are passed as 1-D broadcast terms evaluated in MATLAB's left-to-right order.

precision='single' is the committed revision's typing (s_r = single(linspace..),
:69; single tables, double U_mesh); precision='double' is the revision that
produced test/obj_1.mat (test/test_coder.m:19-36), used for the parity fixture.
"""
from __future__ import annotations

import numpy as np

from .core import Backup
from .matlab_compat import interp_linear_point, linspace
from .problem import ProblemSpec, Term


class Dynamic_Solver:
    def __init__(self, precision="single"):
        # Dynamic_Solver.m:47-64
        self.checkstagesXJF = 1
        self.Q = np.array([[0.25, 0.0], [0.0, 0.05]])
        self.A = np.array([[0.9974, 0.0539], [-0.1078, 1.1591]])
        self.B = np.array([[0.0013], [0.0539]])
        self.R = 0.05
        self.N = 200
        self.S = 2
        self.C = 1
        self.dx = 100
        self.du = 1000
        self.x_max = 3.0
        self.x_min = -2.5
        self.u_max = 10.0
        self.u_min = -40.0
        self.H = None  # unused by the reference too (terminal cost is 0, :83-84)
        self.precision = precision
        self.device = 0
        # results
        self.s_r = None
        self.u_star = None
        self.u_star_idx = None
        self.J_star = None
        self.X1_mesh = None
        self.X2_mesh = None
        self.F_values = None  # obj.F.Values after the sweep (= J at stage 1)
        self.sweep_ms = None
        self.X_path = None
        self.U_path = None

    # ------------------------------------------------------------------
    def build_spec(self):
        """Grid + broadcast terms with MATLAB's typing rules."""
        A, B, Q, R = self.A, self.B, self.Q, self.R
        if self.precision == "single":
            dt = np.float32
            s_r = linspace(self.x_min, self.x_max, self.dx).astype(np.float32)  # :69
            U_mesh = linspace(self.u_min, self.u_max, self.du)                  # :72 (double)
            f = np.float32
            # A(1)*X1 (double scalar * single array -> single), B(1)*U (double) cast on the add
            nxt = [[Term((0,), f(A[0, 0]) * s_r), Term((1,), f(A[0, 1]) * s_r), Term((2,), (B[0, 0] * U_mesh).astype(f))],
                   [Term((0,), f(A[1, 0]) * s_r), Term((1,), f(A[1, 1]) * s_r), Term((2,), (B[1, 0] * U_mesh).astype(f))]]
            cost = [Term((0,), f(Q[0, 0]) * s_r ** 2), Term((1,), f(Q[1, 1]) * s_r ** 2),
                    Term((2,), (R * U_mesh ** 2).astype(f))]  # :198-199 (Q(1), Q(4): diagonal only)
        elif self.precision == "double":
            dt = np.float64
            s_r = linspace(self.x_min, self.x_max, self.dx)                    # test_coder.m:19
            U_mesh = linspace(self.u_min, self.u_max, self.du)
            nxt = [[Term((0,), A[0, 0] * s_r), Term((1,), A[0, 1] * s_r), Term((2,), B[0, 0] * U_mesh)],
                   [Term((0,), A[1, 0] * s_r), Term((1,), A[1, 1] * s_r), Term((2,), B[1, 0] * U_mesh)]]
            cost = [Term((0,), Q[0, 0] * s_r ** 2), Term((1,), Q[1, 1] * s_r ** 2), Term((2,), R * U_mesh ** 2)]
        else:
            raise ValueError("precision must be 'single' or 'double'")
        self.s_r = s_r
        self._U_mesh = U_mesh
        return ProblemSpec([s_r, s_r], [self.du], nxt, cost, dtype=dt, index_base=1)

    def run(self):
        spec = self.build_spec()
        s_r, U_mesh = self.s_r, self._U_mesh
        self.X1_mesh, self.X2_mesh = np.meshgrid(s_r, s_r, indexing="ij")  # ndgrid, :70
        n_st = self.N - 1                                                  # for k=1:N-1, :86
        # Dynamic_Solver.m:212-219 (`checkstagesXJF`, default 1): per stage the reference copies the fixed sub-block
        # (50:55, 52:57, 105) of J_current_state, X_next_M1 and X_next_M2; the library evaluates that block on the GPU
        # (hjb_probe).  MATLAB raises an index error when dx < 57 or du < 105; the mirror leaves the taps None instead
        # (the fixture revision test/test_coder.m has no taps and runs at dx = 35).
        probe = None
        if self.checkstagesXJF and self.dx >= 57 and self.du >= 105:
            probe = {"lo": (49, 51), "hi": (55, 57), "control": (104,), "want": ("g", "x_next", "j_interp")}
        with Backup(spec, device=self.device) as bk:
            out = bk.solve(n_st, keep_J=True, keep_idx=True, probe=probe)
        self.J_current_state_check = self.X_next_M1_check = self.X_next_M2_check = self.J_F_next_check = None
        if probe is not None:
            # plane k_s - 1 holds stage k_s; the reference indexes its taps by the loop counter k = N - k_s
            flip = lambda a: a[..., ::-1]
            self.J_current_state_check = flip(out["probe"]["g"])
            self.X_next_M1_check = flip(out["probe"]["x_next"][:, :, 0, :])
            self.X_next_M2_check = flip(out["probe"]["x_next"][:, :, 1, :])
            self.J_F_next_check = flip(out["probe"]["j_interp"])        # commented out in the reference (:215)
        self.sweep_ms = out["sweep_ms"]
        dt = spec.dtype
        shape = (self.dx, self.dx, self.N)
        # stage k_s lives in column k_s-1; stage N = terminal (zeros)  (:77-80,:100)
        self.J_star = np.zeros(shape, dtype=dt)
        self.u_star = np.zeros(shape, dtype=dt)
        self.J_star[:, :, : n_st] = out["J_stages"].reshape((self.dx, self.dx, n_st), order="F")
        idx = out["idx_stages"].reshape((self.dx, self.dx, n_st), order="F")  # 1-based
        self.u_star[:, :, : n_st] = U_mesh[idx - 1].astype(dt)               # u_star(:,:,k_s) = U_mesh(u_star_idx)
        self.u_star_idx = idx[:, :, 0].copy()      # value left in obj.u_star_idx after the loop (k_s = 1)
        self.F_values = self.J_star[:, :, 0].copy()
        return self

    # ------------------------------------------------------------------
    def a_D(self, X1, X2, Ui):
        # Dynamic_Solver.m:191-194
        return self.A @ np.array([X1, X2], dtype=np.float64) + self.B[:, 0] * Ui

    def get_optimal_path(self, X0=None, mode="Nssu", ssu_num=1):
        """Dynamic_Solver.m:108-181: closed-loop rollout with the stored policy.
        mode 'ssu' freezes the policy of stage `ssu_num` (1-based).  Returns
        (X [S,N], U [N]) instead of plotting."""
        if X0 is None:
            X0, mode, ssu_num = np.array([2.0, 1.0]), "Nssu", 1
        if self.u_star is None:
            raise RuntimeError("run() first")
        N = self.N
        knots = [np.asarray(self.s_r, dtype=np.float64)] * 2
        X = np.zeros((self.S, N))
        U = np.zeros(N)
        X[:, 0] = np.asarray(X0, dtype=np.float64).reshape(-1)
        for k in range(N - 1):
            USM = self.u_star[:, :, ssu_num - 1] if mode == "ssu" else self.u_star[:, :, k]
            U[k] = interp_linear_point(knots, USM, X[:, k])
            X[:, k + 1] = self.a_D(X[0, k], X[1, k], U[k])
        self.X_path, self.U_path = X, U
        if mode == "ssu":
            USTAR_OPT = self.u_star[:, :, 0].astype(np.float64)
            USM = self.u_star[:, :, ssu_num - 1].astype(np.float64)
            self.ssu_tol = float(np.sum(np.sum(USTAR_OPT - USM, axis=0) ** 2))
            self.ssu_err_first = abs(interp_linear_point(knots, USTAR_OPT, X[:, 0]) -
                                     interp_linear_point(knots, USM, X[:, 0]))
        return X, U

    @staticmethod
    def compare_data(obj1, obj2):
        # Dynamic_Solver.m:266-280
        if obj1.J_star is None or obj2.J_star is None or obj1.J_star.size == 0 or obj2.J_star.size == 0:
            raise ValueError("stop throwing empty data at me")
        return bool(np.array_equal(obj1.J_star, obj2.J_star))
