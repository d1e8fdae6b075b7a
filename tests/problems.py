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


def colsweep_problem(seed, n, nU=9, nonuniform=False, gax=3, big=2.7, small=0.6, cost="fast", a1_amp=0.6, levels=5,
                     dtype=np.float32, index_base=1, j_storage=None, a1_axis=None):
    """The shape of the column-sweep stage kernel (variant 7, kernels_colsweep.h; pos-att with the axes relabelled
    (x, theta, v, w)): D = 4, one control dim; axes 0 and 1 move with the state only (axis 0 over dims {0,2,3}, axis 1
    over dims {1,2,3}); axes 2 and 3 move with the control - the "group" axis `gax` by `big` cells times one of
    `levels` distinct values, the other ("window") axis by less than `small` < 1 cells.  a1_amp > 1 makes the axis-1
    cell jump irregularly along a column (re-priming path).  cost: 'fast' (state terms, the dim-1 term last, one
    control term), 'step01' (first term over dims (0,1): nothing is column-invariant, per-step term not uniform),
    'multi' (two control-only terms), 'ctrl_only' (no state term at all).  a1_axis=d: axis 1 moves with dim 1 and
    dim d only, as pos-att's theta+ = theta + h w does (the cooperative form wants d = the group axis the plan picks)."""
    rng = np.random.default_rng(seed)
    assert len(n) == 4
    knots = []
    for a in range(4):
        if nonuniform:
            k = np.cumsum(rng.uniform(0.6, 1.4, n[a]))
            k = (k - k[0]) / (k[-1] - k[0]) * 2.0 - 1.0
        else:
            k = np.linspace(-1.0, 1.0, n[a])
        knots.append(k.astype(dtype).astype(np.float64))
    hs = [2.0 / (n[a] - 1) for a in range(4)]
    wax = 5 - gax
    lev = np.linspace(-1.0, 1.0, levels) if levels > 1 else np.zeros(1)
    d = lev[rng.permutation(nU) % levels]                       # group-axis displacement level of each control
    c = rng.uniform(-1.0, 1.0, nU)                              # window-axis displacement of each control
    nxt = [None] * 4
    nxt[0] = [Term((0,), knots[0].copy()), Term((2,), 0.7 * hs[0] * rng.uniform(-1, 1, n[2])),
              Term((3,), 0.25 * hs[0] * rng.uniform(-1, 1, n[3]))]
    nxt[1] = [Term((1,), knots[1].copy()), Term((3,), a1_amp * hs[1] * rng.uniform(-1, 1, n[3])),
              Term((2,), 0.2 * hs[1] * rng.uniform(-1, 1, n[2]))]
    if a1_axis is not None:
        nxt[1] = [Term((1,), knots[1].copy()), Term((a1_axis,), a1_amp * hs[1] * rng.uniform(-1, 1, n[a1_axis]))]
    nxt[gax] = [Term((gax,), knots[gax].copy()), Term((4,), big * hs[gax] * d)]
    nxt[wax] = [Term((wax,), knots[wax].copy()), Term((4,), small * hs[wax] * c)]
    cu = 0.3 * np.round(rng.uniform(0, 3, nU)) ** 2             # repeated values: exact ties between controls
    st = {a: Term((a,), (1.0 + a) * knots[a] ** 2) for a in range(4)}
    if cost == "fast":
        ct = [st[0], st[2], st[3], st[1], Term((4,), cu)]
    elif cost == "step01":
        ct = [Term((0, 1), 0.1 * rng.random((n[0], n[1]))), st[2], st[3], Term((4,), cu)]
    elif cost == "multi":
        ct = [st[0], st[1], st[3], Term((4,), cu), Term((4,), 0.1 * rng.random(nU))]
    elif cost == "ctrl_only":
        ct = [Term((4,), cu)]
    else:
        raise ValueError(cost)
    return ProblemSpec(knots, [nU], nxt, ct, dtype=dtype, index_base=index_base, j_storage=j_storage)


def rate_shared_problem(seed, n_so, n_rates, m=(11, 11, 11), gain=(0.55, 0.5, 0.6), nonuniform=False, so_move=0.7,
                        dtype=np.float32, index_base=1, j_storage=None):
    """The shape of K15 (kernels_uniwin.h; Solver_attitude.run with the angle axes first): the state-only axes `n_so`, whose
    next value is tabulated over ALL state dims, then three 'rate' axes whose next value depends on the three rates and on
    ONE control dim each (control dim c drives rate axis c; the innermost control the LAST axis).  gain[c] = the move of rate
    axis c, in cells, from the middle to either end of its control's range; costs repeat values (exact ties)."""
    rng = np.random.default_rng(seed)
    NP = len(n_so)
    n = tuple(n_so) + tuple(n_rates)
    D = len(n)
    knots = []
    for a in range(D):
        if nonuniform:
            k = np.cumsum(rng.uniform(0.6, 1.4, n[a]))
            k = (k - k[0]) / (k[-1] - k[0]) * 2.0 - 1.0
        else:
            k = np.linspace(-1.0, 1.0, n[a])
        knots.append(k.astype(dtype).astype(np.float64))
    hs = [2.0 / (n[a] - 1) for a in range(D)]
    all_dims = tuple(range(D))
    nxt = []
    for a in range(NP):
        shape = [1] * D
        shape[a] = n[a]
        tab = knots[a].reshape(shape) + so_move * hs[a] * rng.uniform(-1.0, 1.0, n)
        nxt.append([Term(all_dims, tab)])
    for c in range(3):
        a = NP + c
        others = [NP + x for x in range(3) if x != c]
        dims = tuple(others) + (D + c,)
        tab = 0.08 * hs[a] * rng.standard_normal((n[others[0]], n[others[1]]))[:, :, None] + \
            gain[c] * hs[a] * np.linspace(-1.0, 1.0, m[c])[None, None, :]
        nxt.append([Term((a,), knots[a].copy()), Term(dims, tab)])
    cost = [Term((NP + c,), (1.0 + c) * knots[NP + c] ** 2) for c in range(3)]
    cost.append(Term(tuple(range(NP)), 2.0 * rng.random(tuple(n_so))))
    for c in range(3):
        cost.append(Term((D + c,), 0.25 * np.round(rng.uniform(0, 3, m[c])) ** 2))
    return ProblemSpec(knots, list(m), nxt, cost, dtype=dtype, index_base=index_base, j_storage=j_storage)
