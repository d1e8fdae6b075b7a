"""Seeded synthetic problems for parity tests (not reference code: generators of
inputs).  Every generator returns an hjbdp.ProblemSpec."""
from __future__ import annotations

import numpy as np

from hjbdp import ProblemSpec, Term


def random_problem(seed, n, m, dtype=np.float64, nonuniform=False, spread=0.35, index_base=0):
    """D = len(n) state dims, C = len(m) control dims.  x_next_a = x_a + (terms
    mixing 1..3 other dims and controls); cost = sum of per-dim quadratics + a
    mixed state/control term; queries land inside, on and outside the grid."""
    rng = np.random.default_rng(seed)
    D, C = len(n), len(m)
    g = tuple(n) + tuple(m)
    knots = []
    for a in range(D):
        if nonuniform:
            k = np.cumsum(rng.uniform(0.5, 1.5, n[a]))
            k = (k - k[0]) / (k[-1] - k[0]) * 2.0 - 1.0
        else:
            k = np.linspace(-1.0, 1.0, n[a])
        knots.append(k.astype(dtype).astype(np.float64))
    nxt = []
    for a in range(D):
        terms = [Term((a,), knots[a].copy())]
        # a state-only coupling term over up to 2 other dims
        others = [d for d in range(D) if d != a]
        if others:
            pick = sorted(rng.choice(others, size=min(len(others), int(rng.integers(1, 3))), replace=False).tolist())
            shape = tuple(g[d] for d in pick)
            terms.append(Term(pick, spread * 0.5 * rng.standard_normal(shape)))
        # a control term (some axes have none, some mix a state dim in)
        r = rng.random()
        if r < 0.75:
            c = int(rng.integers(0, C))
            if r < 0.25 and D > 1:
                d = int(rng.choice(others))
                dims = (d, D + c)
                terms.append(Term(dims, spread * rng.standard_normal((g[d], g[D + c]))))
            else:
                terms.append(Term((D + c,), spread * rng.standard_normal(g[D + c])))
        if rng.random() < 0.3:  # a trailing state-only term after a control term
            terms.append(Term((a,), 0.05 * rng.standard_normal(g[a])))
        nxt.append(terms)
    cost = [Term((a,), (1.0 + a) * knots[a] ** 2) for a in range(D)]
    for c in range(C):
        cost.append(Term((D + c,), 0.3 * rng.standard_normal(m[c]) ** 2))
    if D + C <= 5:
        cost.append(Term((0, D), 0.1 * rng.random((g[0], g[D]))))
    return ProblemSpec(knots, m, nxt, cost[:12], dtype=dtype, index_base=index_base)


def random_terminal(spec, seed=0):
    rng = np.random.default_rng(seed)
    return rng.random(spec.nS).astype(spec.dtype)


def nested_problem(seed, n, m, dtype=np.float32, nonuniform=False, spread=0.2, mixed_inner=False, monotone=None):
    """Structure of the spacecraft solvers: control dim c drives state axis
    D-C+c only (the innermost control dim drives the LAST axis); every axis also
    couples to other state dims.  mixed_inner adds an inner term that also
    depends on a state dim (a 'general' inner term for the nested kernel)."""
    rng = np.random.default_rng(seed)
    D, C = len(n), len(m)
    g = tuple(n) + tuple(m)
    knots = []
    for a in range(D):
        if nonuniform:
            k = np.cumsum(rng.uniform(0.5, 1.5, n[a]))
            k = (k - k[0]) / (k[-1] - k[0]) * 2.0 - 1.0
        else:
            k = np.linspace(-1.0, 1.0, n[a])
        knots.append(k.astype(dtype).astype(np.float64))
    nxt = []
    for a in range(D):
        terms = [Term((a,), knots[a].copy())]
        others = [d for d in range(D) if d != a]
        if others:
            pick = sorted(rng.choice(others, size=min(len(others), 2), replace=False).tolist())
            terms.append(Term(pick, spread * 0.5 * rng.standard_normal(tuple(g[d] for d in pick))))
        c = a - (D - C)
        if c >= 0:
            if mixed_inner and others:
                d = int(others[0])
                dims = tuple(sorted((d, D + c)))
                terms.append(Term(dims, spread * rng.standard_normal(tuple(g[x] for x in dims))))
                if mixed_inner == "only" and c == C - 1:     # the inner term depends on a state dim AND the control
                    nxt.append(terms)
                    continue
            tab = spread * rng.standard_normal(g[D + c])
            if monotone and c == C - 1:       # monotone inner control table (what real actuator levels are)
                tab = np.sort(tab) if monotone == "inc" else np.sort(tab)[::-1].copy()
            terms.append(Term((D + c,), tab))
        nxt.append(terms)
    cost = [Term((a,), (1.0 + a) * knots[a] ** 2) for a in range(D)]
    for c in range(C):
        cost.append(Term((D + c,), 0.3 * rng.standard_normal(m[c]) ** 2))
    return ProblemSpec(knots, m, nxt, cost[:12], dtype=dtype, index_base=1)
