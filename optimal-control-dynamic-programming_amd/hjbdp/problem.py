"""Problem description handed to libhjbdp: grid vectors + ordered broadcast terms.

The reference builds its next-state and stage-cost tables by MATLAB implicit
expansion of vectors reshaped onto dims 1..D (states) and D+1..D+C (controls)
(attitude-control/Solver_attitude.m:717-742 `reshape_states`,
pos-att/Solver_pos_att.m:307-314 and :791-801).  A `Term` is one such reshaped
operand; a quantity is the left-to-right sum of its terms, which is how MATLAB
evaluates `A(1)*X1 + A(3)*X2 + B(1)*U` (test/Dynamic_Solver.m:186) element by
element - so nothing of size nS x nU is ever materialised.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi


class Term:
    """`data` varies along the grid dims listed in `dims` (0-based, increasing:
    state dims 0..D-1, then control dims D..D+C-1); data.shape follows dims."""

    def __init__(self, dims, data):
        self.dims = tuple(int(d) for d in dims)
        if list(self.dims) != sorted(set(self.dims)):
            raise ValueError("Term dims must be strictly increasing")
        self.data = np.asarray(data)
        if self.data.ndim != len(self.dims):
            raise ValueError("Term data rank %d != number of dims %d" % (self.data.ndim, len(self.dims)))

    @property
    def mask(self):
        m = 0
        for d in self.dims:
            m |= 1 << d
        return m


class ProblemSpec:
    """knots[D] grid vectors (float64 values, rounded to `dtype` by the library -
    pass `single(linspace(..))`-rounded values to mirror test/Dynamic_Solver.m:69),
    m[C] control grid sizes, next_terms[D][*], cost_terms[*].

    idx_dtype: storage type of the argmin labels - None / np.int32 (default), np.uint8, np.uint16 or "auto" (the
    narrowest that holds nU - 1 + index_base; hjbdp.h hjb_problem.idx_dtype).
    table_dtype: None, or np.float64 with float32 arithmetic = the reference's pos-att typing
    (Solver_pos_att.m:299-327: double query tables, single F_gI.Values): next_terms are kept in float64, each query
    is formed, located and weighted in double and the weight rounded to float32 once (hjbdp.h HJB_TAB_F64).
    cost_dtype: None, or np.float64 with float32 arithmetic: cost_terms are kept in float64 and the stage cost of a
    (state, control) is their ordered sum in double rounded to float32 once - `single(double expression)` of
    Solver_pos_att.m:800-801 without the nS x nU array (hjbdp.h HJB_COST_F64)."""

    def __init__(self, knots, m, next_terms, cost_terms, dtype=np.float64, index_base=0, j_storage=None, model=None,
                 idx_dtype=None, table_dtype=None, cost_dtype=None):
        self.dtype = np.dtype(dtype)
        self.cost_dtype = None if cost_dtype is None else np.dtype(cost_dtype)
        if self.cost_dtype is not None and not (self.cost_dtype == np.float64 and self.dtype == np.float32 and model is None):
            raise ValueError("cost_dtype must be float64 with float32 arithmetic (and no state model)")
        self.table_dtype = None if table_dtype is None else np.dtype(table_dtype)
        if self.table_dtype is not None and not (self.table_dtype == np.float64 and self.dtype == np.float32 and model is None):
            raise ValueError("table_dtype must be float64 with float32 arithmetic (and no state model)")
        if idx_dtype is None or idx_dtype == "auto":
            self.idx_dtype = idx_dtype
        else:
            self.idx_dtype = np.dtype(idx_dtype)
            if self.idx_dtype not in (np.dtype(np.int32), np.dtype(np.uint8), np.dtype(np.uint16)):
                raise ValueError("idx_dtype must be int32, uint8, uint16 or 'auto'")
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        # j_storage=np.float16: J buffers are IEEE half (float32 arithmetic) - HJB_F16S
        self.j_dtype = self.dtype if j_storage is None else np.dtype(j_storage)
        if self.j_dtype != self.dtype and not (self.j_dtype == np.float16 and self.dtype == np.float32):
            raise ValueError("j_storage must be float16 with float32 arithmetic")
        self.knots = [np.ascontiguousarray(k, dtype=np.float64) for k in knots]
        self.D = len(self.knots)
        self.n = tuple(len(k) for k in self.knots)
        self.m = tuple(int(x) for x in m)
        self.C = len(self.m)
        self.G = self.D + self.C
        if not (1 <= self.D <= _abi.HJB_MAX_D):
            raise ValueError("D=%d not in 1..%d" % (self.D, _abi.HJB_MAX_D))
        if not (1 <= self.C <= _abi.HJB_MAX_C):
            raise ValueError("C=%d not in 1..%d" % (self.C, _abi.HJB_MAX_C))
        g = self.n + self.m
        self.next_terms = [[self._check(t, g, self.table_dtype or self.dtype) for t in ts] for ts in next_terms]
        self.cost_terms = [self._check(t, g, self.cost_dtype or self.dtype) for t in cost_terms]
        if len(self.next_terms) != self.D:
            raise ValueError("need one term list per state axis")
        # model = {"kind": "quat_euler321", "h": step, "tables": [x4, x5, x6, x7] over (n0, n1, n2)}: the next
        # value of axes 0..2 is computed inside the library (hjbdp.h HJB_MODEL_QUAT_EULER321), no terms for them
        self.model = None
        model_axes = ()
        if model is not None:
            if model.get("kind") != "quat_euler321":
                raise ValueError("unknown model kind %r" % (model.get("kind"),))
            if self.D != 6 or self.C != 3 or self.dtype != np.float32:
                raise ValueError("quat_euler321 needs D=6, C=3, float32")
            tabs = [np.asfortranarray(t, dtype=np.float32) for t in model["tables"]]
            if len(tabs) != 4 or any(t.shape != self.n[:3] for t in tabs):
                raise ValueError("model tables must be 4 arrays over the first three axes")
            self.model = {"kind": "quat_euler321", "h": float(model["h"]), "tables": tabs}
            model_axes = (0, 1, 2)
        for a, ts in enumerate(self.next_terms + [self.cost_terms]):
            lo = 0 if a in model_axes else 1
            if not (lo <= len(ts) <= (0 if a in model_axes else _abi.HJB_MAX_TERMS)):
                raise ValueError("1..%d terms per quantity (none for model axes)" % _abi.HJB_MAX_TERMS)
        self.index_base = int(index_base)
        self.nS = int(np.prod(self.n))
        self.nU = int(np.prod(self.m))

    @property
    def idx_np_dtype(self):
        """numpy dtype of the labels the library writes for this spec."""
        if self.idx_dtype is None:
            return np.dtype(np.int32)
        if self.idx_dtype == "auto":
            top = self.nU - 1 + self.index_base
            return np.dtype(np.uint8 if top <= 255 else (np.uint16 if top <= 65535 else np.int32))
        return self.idx_dtype

    def _check(self, t, g, dtype=None):
        for ax, d in enumerate(t.dims):
            if not (0 <= d < self.G) or t.data.shape[ax] != g[d]:
                raise ValueError("term over dims %s has shape %s, grid is %s" % (t.dims, t.data.shape, g))
        # column-major bytes in the working dtype
        return Term(t.dims, np.asfortranarray(t.data, dtype=dtype or self.dtype))

    def to_c(self, slab=None):
        """-> (hjb_problem, keepalive list).  slab = (begin, end, halo_lo, halo_hi)."""
        p = _abi.hjb_problem()
        keep = []
        p.D, p.C = self.D, self.C
        for a in range(self.D):
            p.n[a] = self.n[a]
            p.knots[a] = self.knots[a].ctypes.data_as(C.POINTER(C.c_double))
            p.n_next_terms[a] = len(self.next_terms[a])
            for k, t in enumerate(self.next_terms[a]):
                p.next_terms[a][k].mask = t.mask
                p.next_terms[a][k].data = t.data.ctypes.data
                keep.append(t.data)
        for c in range(self.C):
            p.m[c] = self.m[c]
        p.dtype = (_abi.HJB_F16S if self.j_dtype == np.float16 else _abi.HJB_F32) if self.dtype == np.float32 else _abi.HJB_F64
        p.index_base = self.index_base
        p.idx_dtype = {None: _abi.HJB_IDX_I32, "auto": _abi.HJB_IDX_AUTO, np.dtype(np.int32): _abi.HJB_IDX_I32,
                       np.dtype(np.uint8): _abi.HJB_IDX_U8, np.dtype(np.uint16): _abi.HJB_IDX_U16}[self.idx_dtype]
        p.table_dtype = _abi.HJB_TAB_F64 if self.table_dtype is not None else _abi.HJB_TAB_DEFAULT
        p.cost_dtype = _abi.HJB_COST_F64 if self.cost_dtype is not None else _abi.HJB_COST_DEFAULT
        p.n_cost_terms = len(self.cost_terms)
        for k, t in enumerate(self.cost_terms):
            p.cost_terms[k].mask = t.mask
            p.cost_terms[k].data = t.data.ctypes.data
            keep.append(t.data)
        if slab is not None:
            p.slab_begin, p.slab_end, p.halo_lo, p.halo_hi = (int(x) for x in slab)
        if self.model is not None:
            p.model = _abi.HJB_MODEL_QUAT_EULER321
            p.model_h = self.model["h"]
            for i, t in enumerate(self.model["tables"]):
                p.model_tables[i] = t.ctypes.data
                keep.append(t)
        keep.append(self.knots)
        return p, keep


def permute_state_axes(spec: ProblemSpec, order):
    """Relabel the state axes: new axis i = old axis order[i].  Pure bookkeeping (which
    axis is "last" decides which stage kernel applies and the order of the 1-D lerps);
    returns (new_spec, to_old) where to_old(array_flat_F) maps a result laid out on
    the new grid back to the old grid's column-major order."""
    order = tuple(int(a) for a in order)
    D, C = spec.D, spec.C
    if spec.model is not None:
        raise ValueError("a spec with a state model has a fixed axis labelling")
    if sorted(order) != list(range(D)):
        raise ValueError("order must be a permutation of the state axes")
    new_of_old = {old: new for new, old in enumerate(order)}
    for c in range(C):
        new_of_old[D + c] = D + c

    def remap(t):
        nd = [new_of_old[d] for d in t.dims]
        perm = sorted(range(len(nd)), key=lambda i: nd[i])
        return Term(tuple(nd[i] for i in perm), np.transpose(t.data, perm))
    knots = [spec.knots[a] for a in order]
    nxt = [[remap(t) for t in spec.next_terms[a]] for a in order]
    cost = [remap(t) for t in spec.cost_terms]
    new = ProblemSpec(knots, spec.m, nxt, cost, dtype=spec.dtype, index_base=spec.index_base,
                      j_storage=None if spec.j_dtype == spec.dtype else spec.j_dtype,
                      idx_dtype=spec.idx_dtype, table_dtype=spec.table_dtype, cost_dtype=spec.cost_dtype)
    inv = [order.index(a) for a in range(D)]

    def to_old(flat):
        arr = np.asarray(flat).reshape(new.n, order="F")
        return np.transpose(arr, inv).reshape(-1, order="F")
    return new, to_old
