"""Scalar host-side orbital mechanics used by the spacecraft solvers' closed-loop rollouts.

The reference keeps identical copies of seven textbook routines (H. D. Curtis, "Orbital Mechanics for Engineering
Students": universal-variable Kepler propagation, Lagrange f and g, state vector from classical elements, and an
adaptive Runge-Kutta-Fehlberg 4(5) integrator) under position-control/private and pos-att/private.  They run a few
thousand scalar steps after the sweep - sequential, tiny, not a GPU job - and are restated here in plain Python
from the algorithms they implement, keeping the reference's constants and control flow (tolerances, iteration
limits, the integrator's step-size rule) so that a rollout takes the same steps.

No reference artefact pins these ("parity unpinned"): tests/test_host_solvers.py checks them against closed forms
(orbit period, conserved energy and angular momentum, an analytic ODE) and checks the relative-motion equations
against the difference of two independently propagated Kepler orbits.
"""
from __future__ import annotations

import math

import numpy as np

MU_EARTH = 398600.0          # km^3/s^2   (Solver_position.m:192 `mu = 398600`)
R_EARTH = 6378.0             # km         (Solver_position.m:314)


def stumpff_c(z):
    """C(z), position-control/private/stumpC.m (Curtis Eq. 3.53)."""
    if z > 0:
        return (1.0 - math.cos(math.sqrt(z))) / z
    if z < 0:
        return (math.cosh(math.sqrt(-z)) - 1.0) / (-z)
    return 0.5


def stumpff_s(z):
    """S(z), position-control/private/stumpS.m (Curtis Eq. 3.52)."""
    if z > 0:
        r = math.sqrt(z)
        return (r - math.sin(r)) / r ** 3
    if z < 0:
        r = math.sqrt(-z)
        return (math.sinh(r) - r) / r ** 3
    return 1.0 / 6.0


def kepler_universal(dt, r0, vr0, alpha, mu=MU_EARTH, tol=1.0e-8, n_max=1000):
    """Universal anomaly after `dt` by Newton's method (private/kepler_U.m, Curtis Algorithm 3.3):
    same starting guess sqrt(mu)*|alpha|*dt, same stopping rule |F/F'| <= 1e-8, same iteration cap."""
    sq = math.sqrt(mu)
    x = sq * abs(alpha) * dt
    ratio, n = 1.0, 0
    while abs(ratio) > tol and n <= n_max:
        n += 1
        z = alpha * x * x
        c, s = stumpff_c(z), stumpff_s(z)
        f = r0 * vr0 / sq * x * x * c + (1.0 - alpha * r0) * x ** 3 * s + r0 * x - sq * dt
        df = r0 * vr0 / sq * x * (1.0 - z * s) + (1.0 - alpha * r0) * x * x * c + r0
        ratio = f / df
        x -= ratio
    return x


def propagate_kepler(R0, V0, t, mu=MU_EARTH):
    """State (R, V) a time t after (R0, V0) on the two-body orbit: Solver_position.m:333-361 `update_RV_target`
    (Curtis Algorithm 3.4) with the Lagrange coefficients of private/f_and_g.m and private/fDot_and_gDot.m."""
    R0, V0 = np.asarray(R0, dtype=np.float64), np.asarray(V0, dtype=np.float64)
    r0 = math.sqrt(float(R0 @ R0))
    v0 = math.sqrt(float(V0 @ V0))
    vr0 = float(R0 @ V0) / r0
    alpha = 2.0 / r0 - v0 * v0 / mu
    x = kepler_universal(t, r0, vr0, alpha, mu)
    z = alpha * x * x
    f = 1.0 - x * x / r0 * stumpff_c(z)
    g = t - x ** 3 * stumpff_s(z) / math.sqrt(mu)
    R = f * R0 + g * V0
    r = math.sqrt(float(R @ R))
    fdot = math.sqrt(mu) / r / r0 * (z * stumpff_s(z) - 1.0) * x
    gdot = 1.0 - x * x / r * stumpff_c(z)
    return R, fdot * R0 + gdot * V0


def state_from_elements(h, e, raan, incl, argp, theta, mu=MU_EARTH):
    """(r, v) in the geocentric equatorial frame from [h e RA incl w TA]: private/sv_from_coe.m (Curtis
    Algorithm 4.5): perifocal state rotated by (R3(w) R1(i) R3(RA))^T."""
    rp = (h * h / mu) / (1.0 + e * math.cos(theta)) * np.array([math.cos(theta), math.sin(theta), 0.0])
    vp = (mu / h) * np.array([-math.sin(theta), e + math.cos(theta), 0.0])

    def r3(a):
        return np.array([[math.cos(a), math.sin(a), 0.0], [-math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])

    def r1(a):
        return np.array([[1.0, 0.0, 0.0], [0.0, math.cos(a), math.sin(a)], [0.0, -math.sin(a), math.cos(a)]])
    q = (r3(argp) @ r1(incl) @ r3(raan)).T
    return q @ rp, q @ vp


# Fehlberg's 4(5) tableau
_A = (0.0, 1.0 / 4, 3.0 / 8, 12.0 / 13, 1.0, 1.0 / 2)
_B = ((),
      (1.0 / 4,),
      (3.0 / 32, 9.0 / 32),
      (1932.0 / 2197, -7200.0 / 2197, 7296.0 / 2197),
      (439.0 / 216, -8.0, 3680.0 / 513, -845.0 / 4104),
      (-8.0 / 27, 2.0, -3544.0 / 2565, 1859.0 / 4104, -11.0 / 40))
_C4 = np.array([25.0 / 216, 0.0, 1408.0 / 2565, 2197.0 / 4104, -1.0 / 5, 0.0])
_C5 = np.array([16.0 / 135, 0.0, 6656.0 / 12825, 28561.0 / 56430, -9.0 / 50, 2.0 / 55])


def rkf45(rates, t0, tf, y0, tol=1.0e-8):
    """Adaptive RKF4(5) from t0 to tf (private/rkf45.m).  Step-size control as in the reference: first step
    (tf-t0)/100; a step is accepted when the largest 4th-vs-5th order difference is within tol*max(|y|_max, 1);
    the next step is min(delta, 4) times the last with delta = (allowed/(error + eps))^(1/5); an accepted step is
    clipped to the end of the interval AFTER its stage derivatives were formed with the unclipped step (the
    reference's order of operations, kept).  Returns the state at tf (the reference's yout(end,:))."""
    eps = np.finfo(np.float64).eps
    t, y = float(t0), np.asarray(y0, dtype=np.float64).copy()
    h = (tf - t0) / 100.0
    f = np.zeros((y.size, 6))
    while t < tf:
        hmin = 16.0 * float(np.spacing(abs(t)))          # 16*eps(t)
        ti, yi = t, y
        for i in range(6):
            yin = yi.copy()
            for j in range(i):
                yin = yin + h * _B[i][j] * f[:, j]
            f[:, i] = rates(ti + _A[i] * h, yin)
        te_max = float(np.max(np.abs(h * (f @ (_C4 - _C5)))))
        te_allowed = tol * max(float(np.max(np.abs(y))), 1.0)
        delta = (te_allowed / (te_max + eps)) ** 0.2
        if te_max <= te_allowed:
            h = min(h, tf - t)
            t = t + h
            y = yi + h * (f @ _C5)
        h = min(delta * h, 4.0 * h)
        if h < hmin:
            break                        # the reference prints a warning and returns what it has
    return y


def relative_motion_rates(t, y, R0, V0, accel, mu=MU_EARTH):
    """d/dt of the chaser's relative state [dx dy dz dvx dvy dvz] in the target's co-moving (LVLH) frame,
    Solver_position.m:261-309 (Curtis Eq. 7.36): the target state is re-propagated from (R0, V0) to time t; the
    commanded accelerations `accel` = (a_x, a_y, a_z) enter additively."""
    R, V = propagate_kepler(R0, V0, t, mu)
    nr = math.sqrt(float(R @ R))
    rdotv = float(R @ V)
    c = np.cross(R, V)
    H = math.sqrt(float(c @ c))
    dx, dy, dz, dvx, dvy, dvz = (float(v) for v in y)
    dax = (2.0 * mu / nr ** 3 + H * H / nr ** 4) * dx - 2.0 * rdotv / nr ** 4 * H * dy + 2.0 * H / nr ** 2 * dvy + accel[0]
    day = -(mu / nr ** 3 - H * H / nr ** 4) * dy + 2.0 * rdotv / nr ** 4 * H * dx - 2.0 * H / nr ** 2 * dvx + accel[1]
    daz = -mu / nr ** 3 * dz + accel[2]
    return np.array([dvx, dvy, dvz, dax, day, daz])
