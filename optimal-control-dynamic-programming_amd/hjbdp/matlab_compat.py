"""Bit-level restatements of the few MATLAB builtins the reference's grid setup
relies on (so the tables handed to the GPU equal the ones MATLAB would build)."""
from __future__ import annotations

import numpy as np


def linspace(a, b, n):
    """MATLAB linspace: y(i) = a + (i*(b-a))/(n-1), end points forced exact.
    (numpy.linspace multiplies by a precomputed step and differs by <= 1 ulp on
    some entries; this form reproduces test/obj_1.mat's X1_mesh bit for bit -
    checked in tests/golden/make_golden.py.)"""
    a, b, n = float(a), float(b), int(n)
    if n < 1:
        return np.zeros(0)
    if n == 1:
        return np.array([b])
    y = a + (np.arange(n, dtype=np.float64) * (b - a)) / (n - 1)
    y[0], y[-1] = a, b
    return y


def sym_linspace_position(a, b, n):
    """position-control/Solver_position.m:363-371: 2*ceil(n/2)+1 points."""
    if a > 0:
        raise ValueError("minimum states are not negative, use normal linspace")
    h = int(np.ceil(n / 2)) + 1
    v1 = linspace(a, 0.0, h)
    v2 = linspace(0.0, b, h)[1:]
    return np.concatenate([v1, v2])


def sym_linspace_pos_att(a, b, n):
    """pos-att/Solver_pos_att.m:906-918: exactly n points; for even n the
    negative side has n/2 intervals and the positive side n/2-1 -> non-uniform."""
    if a > 0:
        raise ValueError("minimum states are not negative, use normal linspace")
    c = int(np.ceil(n / 2))
    v1 = linspace(a, 0.0, c + 1) if n % 2 == 0 else linspace(a, 0.0, c)
    v2 = linspace(0.0, b, c)[1:]
    return np.concatenate([v1, v2])


def deg2rad(x):
    return np.asarray(x, dtype=np.float64) * (np.pi / 180.0)


def interp_linear_point(knots, V, x):
    """griddedInterpolant(..., V, 'linear') evaluated at ONE point (scalar host
    work of the forward rollouts, e.g. test/Dynamic_Solver.m:132-135): N-linear
    with linear extrapolation."""
    D = len(knots)
    idx, w = [], []
    for a in range(D):
        k = knots[a]
        i = int(np.clip(np.searchsorted(k, x[a], side="right") - 1, 0, len(k) - 2))
        idx.append(i)
        w.append((x[a] - k[i]) / (k[i + 1] - k[i]))
    acc = 0.0
    for c in range(1 << D):
        wt = 1.0
        ii = []
        for a in range(D):
            bit = (c >> a) & 1
            wt *= w[a] if bit else (1.0 - w[a])
            ii.append(idx[a] + bit)
        acc += wt * float(V[tuple(ii)])
    return acc


def interp_nearest_point(knots, V, x):
    """griddedInterpolant(..., V, 'nearest') at one point (policy lookups,
    Solver_position.m:144-146, Solver_pos_att.m:851-861).  Ties at cell
    mid-points round toward the upper knot (unpinned by any fixture)."""
    ii = []
    for a in range(len(knots)):
        k = knots[a]
        i = int(np.clip(np.searchsorted(k, x[a], side="right") - 1, 0, len(k) - 2))
        ii.append(i + 1 if (x[a] - k[i]) >= (k[i + 1] - x[a]) else i)
    return V[tuple(ii)]
