"""Closed-loop rollouts of the attitude and pos-att solvers with the policies the sweeps leave.

Host-side, scalar, O(N_stage) - the reference's own forward simulators restated without their plots:

  attitude-control/Solver_attitude.m   spacecraft_dynamics_list :600-620, next_stage_states :670-696,
                                       linear_control_response :508-591, get_optimal_path :744-833,
                                       get_optimal_path_simplified_testode45 :835-925
  pos-att/Solver_pos_att.m             get_thruster_on_off_optimal :404-449, get_optimal_path :452-730,
                                       get_target_R0V0 :734-753, update_RV_target :755-782,
                                       to_Moments_Forces :804-823, ECI2body :825-829, RSW2ECI :831-847

State conventions of the reference: the attitude state is X = [w1 w2 w3 q1 q2 q3 q4] with q4 the scalar part; the
pos-att state is X = [x(3) v(3) q(4) w(3)].  MATLAB's ode45 is Dormand-Prince 5(4) with RelTol 1e-3 / AbsTol 1e-6:
scipy's RK45 with the same tolerances is the same method (step-size control differs in detail, so trajectories agree to
the integrator tolerance, not bit for bit; no reference artefact exists for them - tests pin invariants instead).
"""
from __future__ import annotations

import math

import numpy as np

from .orbit import MU_EARTH, R_EARTH, propagate_kepler, state_from_elements


# ---- rigid body + quaternion kinematics (scalar-last quaternion) -----------------------------------------------------
def quat_rates(q, w):
    """Solver_attitude.m:617-620 / Solver_pos_att.m ode_eq: q_dot for q = [q1 q2 q3 q4], q4 scalar."""
    q1, q2, q3, q4 = q
    w1, w2, w3 = w
    return 0.5 * np.array([w3 * q2 - w2 * q3 + w1 * q4,
                           -w3 * q1 + w1 * q3 + w2 * q4,
                           w2 * q1 - w1 * q2 + w3 * q4,
                           -w1 * q1 - w2 * q2 - w3 * q3])


def rigid_body_rates_full(w, inertia, torque):
    """w_dot = J \\ (U - w x (J w)) with the full inertia matrix (Solver_attitude.m:913, Solver_pos_att.m ode_eq)."""
    w = np.asarray(w, dtype=np.float64)
    return np.linalg.solve(inertia, np.asarray(torque, dtype=np.float64) - np.cross(w, inertia @ w))


def quat_to_yaw_pitch_roll(qs_first):
    """MATLAB quat2angle (default 'ZYX') for a scalar-FIRST quaternion [q0 q1 q2 q3] -> (yaw, pitch, roll)."""
    q0, q1, q2, q3 = qs_first
    yaw = math.atan2(2.0 * (q1 * q2 + q0 * q3), q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3)
    s = -2.0 * (q1 * q3 - q0 * q2)
    pitch = math.asin(max(-1.0, min(1.0, s)))
    roll = math.atan2(2.0 * (q2 * q3 + q0 * q1), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3)
    return yaw, pitch, roll


def angle_to_quat(yaw, pitch, roll):
    """MATLAB angle2quat (default 'ZYX') -> scalar-FIRST quaternion."""
    cy, sy = math.cos(yaw / 2), math.sin(yaw / 2)
    cp, sp = math.cos(pitch / 2), math.sin(pitch / 2)
    cr, sr = math.cos(roll / 2), math.sin(roll / 2)
    return np.array([cy * cp * cr + sy * sp * sr, cy * cp * sr - sy * sp * cr,
                     cy * sp * cr + sy * cp * sr, sy * cp * cr - cy * sp * sr])


def _ode45_step(rates, t0, t1, y0):
    """One [t0, t1] integration the way the reference calls ode45 inside its stage loop."""
    from scipy.integrate import solve_ivp
    sol = solve_ivp(rates, (t0, t1), np.asarray(y0, dtype=np.float64), method="RK45", rtol=1e-3, atol=1e-6)
    if not sol.success:
        raise RuntimeError("ode45 step failed: " + sol.message)
    return sol.y[:, -1]


# ---- Solver_attitude ---------------------------------------------------------------------------------------------------
DEFAULT_X0_ATTITUDE = np.array([0.0, 0.0, 0.0, 0.0501511024391496, 0.0833950587800888, -0.0818761044636256,
                                0.991880252153991])      # Solver_attitude.m:160-164


def spacecraft_dynamics_list(sa, X, U):
    """Solver_attitude.m:600-620: x_dot = f(X, u), diagonal inertia (J1, J2, J3)."""
    x1, x2, x3, x4, x5, x6, x7 = X
    u1, u2, u3 = U
    return np.array([(sa.J2 - sa.J3) / sa.J1 * x2 * x3 + u1 / sa.J1,
                     (sa.J3 - sa.J1) / sa.J2 * x3 * x1 + u2 / sa.J2,
                     (sa.J1 - sa.J2) / sa.J3 * x1 * x2 + u3 / sa.J3,
                     0.5 * (x3 * x5 - x2 * x6 + x1 * x7),
                     0.5 * (-x3 * x4 + x1 * x6 + x2 * x7),
                     0.5 * (x2 * x4 - x1 * x5 + x3 * x7),
                     0.5 * (-x1 * x4 - x2 * x5 - x3 * x6)])


def next_stage_states(sa, X1, U, h, mode="RK4"):
    """Solver_attitude.m:670-696: one step of the 7-state dynamics, then renormalise the quaternion."""
    X1 = np.asarray(X1, dtype=np.float64)
    k1 = spacecraft_dynamics_list(sa, X1, U)
    if mode == "RK4":
        k2 = spacecraft_dynamics_list(sa, X1 + k1 * h / 2, U)
        k3 = spacecraft_dynamics_list(sa, X1 + k2 * h / 2, U)
        k4 = spacecraft_dynamics_list(sa, X1 + k3 * h, U)
        X2 = X1 + h * (k1 + 2 * k2 + 2 * k3 + k4) / 6
    elif mode == "taylor":
        X2 = X1 + h * k1
    else:
        raise ValueError("mode must be 'RK4' or 'taylor'")
    X2[3:7] /= math.sqrt(float(np.sum(X2[3:7] ** 2)))
    return X2


def linear_control_response(sa, X0=None, T_final=None, dt=None):
    """Solver_attitude.m:508-591: the PD reference controller U = -K qe(1:3) - C w (K = 0.2 I, C = I) rolled out with
    RK4 steps.  Returns (X [7, N+1], U [3, N], angles [3, N] = yaw, pitch, roll)."""
    X0 = DEFAULT_X0_ATTITUDE if X0 is None else np.asarray(X0, dtype=np.float64)
    T_final = sa.T_final if T_final is None else T_final
    dt = sa.h if dt is None else dt
    N = int(round(T_final / dt))
    X = np.zeros((7, N + 1))
    U = np.zeros((3, N))
    ang = np.zeros((3, N))
    X[:, 0] = X0
    for k in range(N):
        q, w = X[3:7, k], X[0:3, k]
        U[:, k] = -0.2 * q[0:3] - w
        X[:, k + 1] = next_stage_states(sa, X[:, k], U[:, k], dt)
        ang[:, k] = quat_to_yaw_pitch_roll([X[6, k], X[5, k], X[4, k], X[3, k]])      # quat2angle([X7 X6 X5 X4])
    return X, U, ang


def attitude_optimal_path(sa, X0=None, method="nearest", n_steps=None):
    """Solver_attitude.m:744-833 after `run`: at every stage convert the quaternion to (yaw, pitch, roll), look the
    three torques up in the 6-D policy tables U{1,2,3}_Opt over (w1, w2, w3, yaw, pitch, roll), take one first-order
    ('taylor') step.  Returns (X [7, N], U [3, N], X_ANGLES [9, N])."""
    from .matlab_compat import interp_linear_point, interp_nearest_point
    if sa.U1_Opt is None or np.ndim(sa.U1_Opt) != 6:
        raise RuntimeError("run() first")
    X0 = DEFAULT_X0_ATTITUDE if X0 is None else np.asarray(X0, dtype=np.float64)
    N = sa.N_stage if n_steps is None else min(sa.N_stage, int(n_steps) + 1)
    knots = sa.grid_vectors_full()
    look = interp_nearest_point if method == "nearest" else interp_linear_point
    X = np.zeros((7, N))
    U = np.zeros((3, N))
    XA = np.zeros((9, N))
    X[:, 0] = X0
    for k in range(N - 1):
        yaw, pitch, roll = quat_to_yaw_pitch_roll([X[6, k], X[5, k], X[4, k], X[3, k]])
        p = (X[0, k], X[1, k], X[2, k], yaw, pitch, roll)
        U[:, k] = [float(look(knots, T, p)) for T in (sa.U1_Opt, sa.U2_Opt, sa.U3_Opt)]
        X[:, k + 1] = next_stage_states(sa, X[:, k], U[:, k], sa.h, "taylor")
        XA[:, k] = [X[0, k], X[1, k], X[2, k], math.degrees(roll), math.degrees(pitch), math.degrees(yaw), *U[:, k]]
    return X, U, XA


def attitude_optimal_path_simplified(sa, X0=None, n_steps=None):
    """Solver_attitude.m:835-925 after `simplified_run`: the three 2-D (w_i, theta_i) 'nearest' policies drive the FULL
    rigid body (full inertia matrix), integrated with ode45 over each stage.  Returns (T [N], X [N, 7], U [N, 3])."""
    if sa.U1_Opt is None or not callable(sa.U1_Opt):
        raise RuntimeError("simplified_run() first")
    X0 = DEFAULT_X0_ATTITUDE if X0 is None else np.asarray(X0, dtype=np.float64)
    N = sa.N_stage if n_steps is None else min(sa.N_stage, int(n_steps) + 1)
    X = np.zeros((N, 7))
    U = np.zeros((N, 3))
    X[0] = X0
    for k in range(N - 1):
        xs = X[k]
        u = np.array([float(sa.U1_Opt(xs[0], 2 * math.asin(xs[3]))), float(sa.U2_Opt(xs[1], 2 * math.asin(xs[4]))),
                      float(sa.U3_Opt(xs[2], 2 * math.asin(xs[5])))])
        U[k] = u

        def rates(t, y, u=u):
            return np.concatenate([rigid_body_rates_full(y[0:3], sa.InertiaM, u), quat_rates(y[3:7], y[0:3])])
        X[k + 1] = _ode45_step(rates, k * sa.h, (k + 1) * sa.h, xs)
    return np.arange(N) * sa.h, X, U


# ---- Solver_pos_att ----------------------------------------------------------------------------------------------------
def target_R0V0():
    """Solver_pos_att.m:734-753: the target's initial state (perigee altitude 300 km, e = 0.1, equatorial, at perigee)."""
    rp, e = R_EARTH + 300.0, 0.1
    ra = rp * (1.0 + e) / (1.0 - e)
    h_ = math.sqrt(2.0 * MU_EARTH * rp * ra / (ra + rp))
    return state_from_elements(h_, e, 0.0, 0.0, 0.0, 0.0, MU_EARTH)


def RSW2ECI(pos, vel):
    """Solver_pos_att.m:831-847: columns R, S, W."""
    pos, vel = np.asarray(pos, dtype=np.float64), np.asarray(vel, dtype=np.float64)
    R = pos / np.linalg.norm(pos)
    c = np.cross(pos, vel)
    W = c / np.linalg.norm(c)
    S = np.cross(W, R)
    return np.column_stack([R, S, W])


def ECI2body(q):
    """Solver_pos_att.m:825-829 (scalar-last quaternion)."""
    q1, q2, q3, q4 = q
    return np.array([[1 - 2 * (q2 * q2 + q3 * q3), 2 * (q1 * q2 + q3 * q4), 2 * (q1 * q3 - q2 * q4)],
                     [2 * (q2 * q1 - q3 * q4), 1 - 2 * (q1 * q1 + q3 * q3), 2 * (q2 * q3 + q1 * q4)],
                     [2 * (q3 * q1 + q2 * q4), 2 * (q3 * q2 - q1 * q4), 1 - 2 * (q1 * q1 + q2 * q2)]])


def to_Moments_Forces(pa, f, R0, V0, q):
    """Solver_pos_att.m:804-823: thruster levels f[0..11] -> body moments U_M [x, y, z] and the acceleration in the RSW
    frame (body-frame thrust sums rotated back through ECI2body and RSW2ECI)."""
    f = np.asarray(f, dtype=np.float64)
    U_M = np.array([(f[4] - f[5] + f[10] - f[11]) * pa.T_dist,       # x
                    (f[0] - f[1] + f[6] - f[7]) * pa.T_dist,         # y
                    (f[2] - f[3] + f[8] - f[9]) * pa.T_dist])        # z
    a_body = np.array([f[0] + f[1] + f[6] + f[7], f[2] + f[3] + f[8] + f[9], f[4] + f[5] + f[10] + f[11]]) / pa.Mass
    acc = np.linalg.solve(RSW2ECI(R0, V0), np.linalg.solve(ECI2body(q), a_body))
    return U_M, acc


def thruster_policies(pa):
    """set_controller (:849-884) for the three channels from the controllers simplified_run left in memory:
    12 'nearest' policies over (x, v, theta, w), indexed by thruster number."""
    from .solver_position import NearestPolicy
    names = {"x": (0, 1, 6, 7), "y": (2, 3, 8, 9), "z": (4, 5, 10, 11)}
    pol = [None] * 12
    for ch, thr in names.items():
        c = pa.controllers["channel_%s_controller_1" % ch]
        idx = np.asarray(c["U_Optimal_id"], dtype=np.int64) - 1
        for key, t in zip(("f0_allcomb", "f1_allcomb", "f6_allcomb", "f7_allcomb"), thr):
            pol[t] = NearestPolicy(c["GridVectors"], np.asarray(c[key])[idx])
    return pol


def get_thruster_on_off_optimal(pol, x, v, t, w, R0, V0, q):
    """Solver_pos_att.m:404-449: rotate the relative position / velocity RSW -> ECI -> body, then per channel look up
    its four thrusters at (x_i, v_i, theta, w): channel x uses the angle / rate about y, y about z, z about x."""
    M = ECI2body(q) @ RSW2ECI(R0, V0)
    xb, vb = M @ np.asarray(x, dtype=np.float64), M @ np.asarray(v, dtype=np.float64)
    f = np.zeros(12)
    for (thr, i, ax) in (((0, 1, 6, 7), 0, 1), ((2, 3, 8, 9), 1, 2), ((4, 5, 10, 11), 2, 0)):
        for tn in thr:
            f[tn] = float(pol[tn](xb[i], vb[i], t[ax], w[ax]))
    return f


def pos_att_optimal_path(pa, X0=None, n_steps=None):
    """Solver_pos_att.m:452-730 without the plots: 13-state closed loop.  Per stage: thruster levels from the policies,
    forces / moments held over the stage, ode45 of the relative-motion equations about the Kepler-propagated target
    plus the rigid body with the full inertia matrix.  Returns (T [N], X [N, 13], F_Th_Opt [N, 12], Force_Moment [N, 6])."""
    if not pa.controllers:
        raise RuntimeError("simplified_run() first")
    if X0 is None:
        q0 = angle_to_quat(math.radians(0.0), math.radians(3.0), math.radians(0.0))[::-1]      # :461-462
        X0 = np.concatenate([[-0.1, 0.0, 0.0], [0.0, 0.0, 0.0], q0, [0.0, 0.0, 0.0]])
    pol = thruster_policies(pa)
    N = pa.N_stage if n_steps is None else min(pa.N_stage, int(n_steps) + 1)
    X = np.zeros((N, 13))
    F = np.zeros((N, 12))
    FM = np.zeros((N, 6))
    X[0] = X0
    R0, V0 = target_R0V0()
    mu = MU_EARTH
    for k in range(N - 1):
        xs = X[k]
        t_stage = 2.0 * np.arcsin(xs[6:9])
        f = get_thruster_on_off_optimal(pol, xs[0:3], xs[3:6], t_stage, xs[10:13], R0, V0, xs[6:10])
        U_M, acc = to_Moments_Forces(pa, f, R0, V0, xs[6:10])
        F[k] = f
        FM[k] = np.concatenate([acc, U_M])

        def rates(t, y, acc=acc, U_M=U_M):
            R, V = propagate_kepler(R0, V0, t, mu)
            nR = math.sqrt(float(R @ R))
            RdV = float(R @ V)
            H = float(np.linalg.norm(np.cross(R, V)))
            x1, x2, x3, v1, v2, v3 = y[0:6]
            return np.concatenate([
                [v1, v2, v3,
                 (2 * mu / nR ** 3 + H * H / nR ** 4) * x1 - 2 * RdV / nR ** 4 * H * x2 + 2 * H / nR ** 2 * v2 + acc[0],
                 -(mu / nR ** 3 - H * H / nR ** 4) * x2 + 2 * RdV / nR ** 4 * H * x1 - 2 * H / nR ** 2 * v1 + acc[1],
                 -mu / nR ** 3 * x3 + acc[2]],
                quat_rates(y[6:10], y[10:13]),
                rigid_body_rates_full(y[10:13], pa.InertiaM, U_M)])
        X[k + 1] = _ode45_step(rates, k * pa.h, (k + 1) * pa.h, xs)
    return np.arange(N) * pa.h, X, F, FM
