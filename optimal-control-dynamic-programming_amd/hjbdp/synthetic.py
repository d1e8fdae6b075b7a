"""Synthetic throughput workloads of SURVEY.md 8(d) / BASELINE.json `configs`,
expressed exactly like the reference expresses its own problems (grid vectors +
broadcast terms).  Deterministic, no RNG."""
from __future__ import annotations

import numpy as np

from .matlab_compat import linspace
from .problem import ProblemSpec, Term


def position3d_spec(n=101, mu=21, h=0.005, mass=4.16, qx=6.0, r=0.1, x_lim=0.5, u_lim=0.26, dtype=np.float32,
                    n_last=None, j_storage=None):
    """C2: 'Solver_position 3-DOF, 101^3 state x 21^3 control grid' - the joint 3-D
    generalisation of position-control/Solver_position.m (:49-72 ranges, Mass, Q, R;
    :84 thrust +-0.26).  D=3 affine chain x+ = (I + h*N) x + (h/Mass) u with N
    strictly upper-triangular ones; cost 6|x|^2 + 0.1|u|^2; controls = ndgrid of
    linspace(-0.26,0.26,mu)^3, first factor fastest.  n_last overrides the size of
    the last state axis (weak-scaling runs extend it by the number of GPUs; the
    range grows with it so the spacing, hence the per-cell work, is unchanged)."""
    f = np.dtype(dtype).type
    n3 = int(n_last) if n_last else n
    k = linspace(-x_lim, x_lim, n).astype(dtype)
    step = 2.0 * x_lim / (n - 1)
    k3 = k if n3 == n else (-x_lim + step * np.arange(n3)).astype(dtype)
    u = linspace(-u_lim, u_lim, mu)
    knots = [k, k, k3]
    A = np.eye(3) + h * np.triu(np.ones((3, 3)), 1)
    b = h / mass
    nxt = []
    for a in range(3):
        terms = [Term((j,), f(A[a, j]) * knots[j]) for j in range(3) if A[a, j] != 0.0]
        terms.append(Term((3 + a,), (b * u).astype(dtype)))
        nxt.append(terms)
    cost = [Term((j,), f(qx) * knots[j] ** 2) for j in range(3)]
    cost += [Term((3 + c,), (r * u ** 2).astype(dtype)) for c in range(3)]
    return ProblemSpec(knots, [mu, mu, mu], nxt, cost, dtype=dtype, index_base=1, j_storage=j_storage)
