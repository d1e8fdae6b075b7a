"""CPU oracle (numpy) for the Bellman-backup hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the one statement every solver of the reference
repeats per stage

    [F.Values, idx] = min( J_stage + F(x_next_1, ..., x_next_D), [], ctrl_dim )

  - test/Dynamic_Solver.m:207-210      (Kirk 2-state example, J_state_M)
  - test/test_coder.m:28-36,109-117    (the revision that produced test/obj_1.mat)
  - position-control/Solver_position.m:135-137
  - attitude-control/Solver_attitude.m:239-241 and :400-409 (3-level min cascade)
  - pos-att/Solver_pos_att.m:272

where F is griddedInterpolant(..., 'linear'): N-linear interpolation with LINEAR
EXTRAPOLATION outside the grid (MATLAB's default for Method='linear').

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (the HIP library) never does.

Parity pin: `sweep()` reproduces the reference's only numeric artefact,
test/obj_1.mat (J_star, u_star 35x35x130 float64), see tests/test_oracle_golden.py.
Everything else (D > 2, non-uniform knots, float32 typing) is pinned only by
MathWorks' documented semantics of griddedInterpolant/min - "parity unpinned" for
those, as DESIGN.md states.

Problem representation (shared with the C twin and the HIP library)
-------------------------------------------------------------------
The reference builds its next-state and stage-cost tables by MATLAB implicit
expansion of vectors reshaped onto dims 1..D (states) and D+1..D+C (controls)
(Solver_attitude.m:717-742 reshape_states, Solver_pos_att.m:307-314).  We keep
exactly that: a quantity is an ORDERED SUM of broadcast terms

    q = ((t0 + t1) + t2) + ...

each term an array over a subset ("mask") of the G = D + C grid dims, stored
column-major over its masked dims.  Left-to-right evaluation reproduces MATLAB's
elementwise evaluation order, so no D+C dimensional table is ever materialised.
"""
from __future__ import annotations

import numpy as np

MAX_D = 6
MAX_C = 3


class Term:
    """One broadcast term: `data` is indexed by the grid dims listed in `dims`
    (0-based, increasing; state dims 0..D-1, control dims D..D+C-1)."""

    def __init__(self, dims, data):
        self.dims = tuple(int(d) for d in dims)
        assert list(self.dims) == sorted(set(self.dims)), "dims must be increasing"
        self.data = np.asarray(data)
        if self.data.ndim != len(self.dims):
            raise ValueError("term data rank %d != len(dims) %d" % (self.data.ndim, len(self.dims)))

    def expand(self, G):
        """View with singleton axes inserted so it broadcasts over all G dims."""
        shape = [1] * G
        for ax, d in enumerate(self.dims):
            shape[d] = self.data.shape[ax]
        return self.data.reshape(shape)


class Problem:
    """Grid + next-state terms + stage-cost terms.

    n[D]      state grid sizes         knots[D]   strictly increasing grid vectors
    m[C]      control grid sizes       (C control dims, cascade order = dim order)
    next_terms[a]  ordered terms of x_next_a      cost_terms   ordered terms of g
    dtype     np.float64 / np.float32: the arithmetic type of the whole backup
    """

    def __init__(self, knots, m, next_terms, cost_terms, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.knots = [np.asarray(k, dtype=np.float64).astype(self.dtype) for k in knots]
        self.D = len(self.knots)
        self.n = tuple(len(k) for k in self.knots)
        self.m = tuple(int(x) for x in m)
        self.C = len(self.m)
        self.G = self.D + self.C
        self.g = self.n + self.m
        assert 1 <= self.D <= MAX_D and 1 <= self.C <= MAX_C
        for k in self.knots:
            assert len(k) >= 2 and np.all(np.diff(k) > 0), "knots must be strictly increasing"
        self.next_terms = [[self._chk(t) for t in ts] for ts in next_terms]
        self.cost_terms = [self._chk(t) for t in cost_terms]
        assert len(self.next_terms) == self.D
        self.nS = int(np.prod(self.n))
        self.nU = int(np.prod(self.m))

    def _chk(self, t):
        for ax, d in enumerate(t.dims):
            assert 0 <= d < self.G and t.data.shape[ax] == self.g[d], (t.dims, t.data.shape, self.g)
        return Term(t.dims, np.ascontiguousarray(t.data, dtype=self.dtype))

    # ordered left-to-right sum, MATLAB elementwise order
    def _sum(self, terms):
        acc = None
        for t in terms:
            e = t.expand(self.G)
            acc = e if acc is None else (acc + e).astype(self.dtype, copy=False)
        return acc


def cell_and_weight(knots, q):
    """Per-axis cell index and (unclamped) weight.

    i = clamp(upper_bound(knots, q) - 1, 0, n-2);  t = (q - k[i]) * (1/(k[i+1]-k[i]))
    t is NOT clamped -> linear extrapolation (griddedInterpolant 'linear' default;
    relied on by the reference: test/test_griddedInterp.m:20 queries far outside).
    The reciprocal spacing is formed in the working dtype, as the HIP path does.
    """
    n = len(knots)
    i = np.searchsorted(knots, q, side="right") - 1
    i = np.clip(i, 0, n - 2)
    rdx = (knots.dtype.type(1) / (knots[1:] - knots[:-1])).astype(knots.dtype)
    t = ((q - knots[i]) * rdx[i]).astype(knots.dtype, copy=False)
    return i, t


def interp_linear(knots, J, q):
    """N-linear interpolation/extrapolation of J (shape n, column-major meaning:
    J[i1,...,iD]) at broadcastable query arrays q[a].

    Evaluation order: successive 1-D lerps v0 + t*(v1 - v0), axis 1 first (the
    memory-contiguous axis), then 2, ... D.  MATLAB's internal order is closed
    source; any order agrees to a few ulp (obj_1.mat is matched to ~1e-13).
    """
    D = len(knots)
    dt = J.dtype
    cells, ts = [], []
    for a in range(D):
        i, t = cell_and_weight(knots[a], q[a])
        cells.append(i)
        ts.append(t)
    shape = np.broadcast_shapes(*[c.shape for c in cells])
    # gather the 2^D corners, lerp axis 0 first
    vals = []
    for corner in range(1 << D):
        idx = tuple(np.broadcast_to(cells[a] + ((corner >> a) & 1), shape) for a in range(D))
        vals.append(J[idx])
    for a in range(D):
        t = ts[a]
        nxt = []
        for j in range(0, len(vals), 2):
            v0, v1 = vals[j], vals[j + 1]
            nxt.append((v0 + (t * (v1 - v0)).astype(dt, copy=False)).astype(dt, copy=False))
        vals = nxt
    return vals[0]


def backup_stage(p: Problem, J_next, chunk_states=None):
    """One Bellman backup: returns (J_k [n], idx [n] int32 0-based flat control index).

    The flat control index enumerates the C control dims column-major (first
    control dim fastest) = MATLAB ndgrid order (Solver_pos_att.m:887-891).
    Ties: MATLAB `min` returns the first index; the attitude solver's cascade
    min over dims 9,8,7 (Solver_attitude.m:400-409) prefers the smallest i1, then
    i2, then i3.  With a single control dim both rules coincide.  We implement
    the cascade rule: lexicographic (i1, i2, i3) with i1 most significant.
    """
    J_next = np.asarray(J_next, dtype=p.dtype).reshape(p.n)
    q = [p._sum(p.next_terms[a]) for a in range(p.D)]
    g = p._sum(p.cost_terms)
    full = p.n + p.m
    # evaluate over the full grid (tests keep sizes small)
    Jf = interp_linear(p.knots, J_next, [np.broadcast_to(x, full) for x in q])
    tot = (np.broadcast_to(g, full) + Jf).astype(p.dtype, copy=False)
    # first minimum over a ROW-major flatten of the control dims (i1 slowest)
    # = lexicographically smallest (i1, i2, i3) among joint minimisers = cascade
    tot = np.ascontiguousarray(tot).reshape(p.n + (p.nU,))
    k = np.argmin(tot, axis=-1)
    Jk = np.take_along_axis(tot, k[..., None], axis=-1)[..., 0]
    sub = np.unravel_index(k, p.m)  # (i1, ..., iC), C-order
    flat = np.zeros(p.n, dtype=np.int64)
    mul = 1
    for c in range(p.C):  # column-major flat label (first control dim fastest)
        flat += sub[c].astype(np.int64) * mul
        mul *= p.m[c]
    return Jk, flat.astype(np.int32)


def sweep(p: Problem, n_stages, terminal=None, keep=False, monitor_period=0, monitor_tol=0.0):
    """Backward sweep of `n_stages` backups starting from the terminal cost
    (default 0: Dynamic_Solver.m:83-84, H is unused).

    keep=True returns per-stage J and idx lists ordered as computed (first entry
    = stage N-1, i.e. reference k=1; Dynamic_Solver.m:86-100 writes k_s = N-k).
    monitor: Solver_pos_att.m:268-285 early stop, restated as intended
    (idsum50_prev treated as 0-initialised): with k_s counting down from
    n_stages, when k_s % period == 0, e = sum(J) - previous sum; stop if |e| < tol.
    """
    J = np.zeros(p.n, dtype=p.dtype) if terminal is None else np.asarray(terminal, dtype=p.dtype).reshape(p.n)
    Js, Is = [], []
    idx = None
    fsum_prev = 0.0
    done = 0
    for k_s in range(n_stages, 0, -1):
        J, idx = backup_stage(p, J)
        done += 1
        if keep:
            Js.append(J.copy())
            Is.append(idx.copy())
        if monitor_period and (k_s % monitor_period == 0):
            fsum = float(np.sum(J.astype(np.float64)))
            e = fsum - fsum_prev
            fsum_prev = fsum
            if abs(e) < monitor_tol:
                break
    if keep:
        return J, idx, Js, Is, done
    return J, idx, done
