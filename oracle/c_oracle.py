"""ctypes wrapper of oracle/libhjb_oracle.so (the C twin, oracle/hjb_oracle.c).
TEST INFRASTRUCTURE ONLY - imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.

It takes the same `hjb_problem` struct the HIP library takes (built by
hjbdp.problem.ProblemSpec.to_c), so both sides see byte-identical tables.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SO = HERE / "libhjb_oracle.so"
_lib = None


def build(force=False):
    if force or not SO.exists() or SO.stat().st_mtime < (HERE / "hjb_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "-B" if force else "-s"], check=True, capture_output=True)
    return SO


def lib(abi):
    """abi = the hjbdp._abi module (struct definitions)."""
    global _lib
    if _lib is None:
        if not SO.exists():
            build()
        # HJB_ORACLE_LIB: another build of the same source (tools/sanitize_cpu.sh: -fsanitize=address,undefined)
        l = C.CDLL(os.environ.get("HJB_ORACLE_LIB") or str(SO))
        l.orc_backup_stage.restype = C.c_int
        l.orc_backup_stage.argtypes = [C.POINTER(abi.hjb_problem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        l.orc_backup_stage_avx2.restype = C.c_int
        l.orc_backup_stage_avx2.argtypes = l.orc_backup_stage.argtypes
        l.orc_sweep.restype = C.c_int
        l.orc_sweep.argtypes = [C.POINTER(abi.hjb_problem), C.POINTER(abi.hjb_solve_opts), C.POINTER(abi.hjb_result), C.c_int]
        l.orc_max_threads.restype = C.c_int
        l.orc_lookup.restype = C.c_int
        l.orc_lookup.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_double)), C.c_void_p,
                                 C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        l.orc_backup_states.restype = C.c_int
        l.orc_backup_states.argtypes = [C.POINTER(abi.hjb_problem), C.POINTER(C.c_void_p), C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_int]
        l.orc_backup_states_from_J.restype = C.c_int
        l.orc_backup_states_from_J.argtypes = [C.POINTER(abi.hjb_problem), C.c_void_p, C.c_void_p, C.c_int64,
                                               C.c_void_p, C.c_void_p, C.c_int]
        l.orc_backup_states_touch.restype = C.c_int
        l.orc_backup_states_touch.argtypes = [C.POINTER(abi.hjb_problem), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]
        l.orc_backup_states_sparse.restype = C.c_int
        l.orc_backup_states_sparse.argtypes = [C.POINTER(abi.hjb_problem), C.c_void_p, C.c_void_p, C.c_int64,
                                               C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        l.orc_canon_eval.restype = None
        l.orc_canon_eval.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        l.orc_monitor_sums.restype = C.c_int
        l.orc_monitor_sums.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double)]
        _lib = l
    return _lib


def monitor_sums(abi, spec, J, idx, single=False):
    """(sum of J, sum of labels) as orc_sweep's monitor forms them (single: the stated float32 tree) - for checking the
    library's monitor on a J the GPU produced."""
    l = lib(abi)
    Jc = np.ascontiguousarray(np.asarray(J, dtype=spec.j_dtype).reshape(-1))
    ic = np.ascontiguousarray(np.asarray(idx).reshape(-1), dtype=np.int32)
    out = (C.c_double * 2)()
    l.orc_monitor_sums(int(spec.to_c()[0].dtype), Jc.ctypes.data, ic.ctypes.data, Jc.size, 1 if single else 0, out)
    return float(out[0]), float(out[1])


def backup_states(abi, spec, jsep, states, nthreads=0):
    """Backup of the listed whole-grid states with J_next(i) = ((jsep[0][i0] + jsep[1][i1]) + ...)
    (float32): the checker for grids whose J does not fit in host memory.  -> (J[k], idx[k])."""
    l = lib(abi)
    p, keep = spec.to_c()
    vecs = [np.ascontiguousarray(v, dtype=np.float32) for v in jsep]
    ptrs = (C.c_void_p * len(vecs))(*[v.ctypes.data for v in vecs])
    st_ = np.ascontiguousarray(states, dtype=np.int64)
    J = np.empty(len(st_), dtype=np.float32)
    idx = np.empty(len(st_), dtype=np.int32)
    st = l.orc_backup_states(C.byref(p), ptrs, st_.ctypes.data, len(st_), J.ctypes.data, idx.ctypes.data,
                             nthreads or l.orc_max_threads())
    if st != 0:
        raise RuntimeError("orc_backup_states status %d" % st)
    return J, idx


def backup_states_from_J(abi, spec, J_next, states, nthreads=0):
    """Backup of the listed whole-grid states from a whole-grid float32 J_next held by the host - the deep-sweep check:
    the GPU's own previous stage, downloaded.  -> (J[k], idx[k])."""
    l = lib(abi)
    p, keep = spec.to_c()
    Jn = np.ascontiguousarray(np.asarray(J_next, dtype=np.float32).reshape(-1, order="F"))
    if Jn.size != spec.nS:
        raise ValueError("J_next must hold the whole grid")
    st_ = np.ascontiguousarray(states, dtype=np.int64)
    J = np.empty(len(st_), dtype=np.float32)
    idx = np.empty(len(st_), dtype=np.int32)
    st = l.orc_backup_states_from_J(C.byref(p), Jn.ctypes.data, st_.ctypes.data, len(st_), J.ctypes.data, idx.ctypes.data,
                                    nthreads or l.orc_max_threads())
    if st != 0:
        raise RuntimeError("orc_backup_states_from_J status %d" % st)
    return J, idx


def backup_states_touch(abi, spec, states, nthreads=0):
    """Sorted unique offsets of every J_next element the backups of the listed states read (all controls, all corners)."""
    l = lib(abi)
    p, keep = spec.to_c()
    st_ = np.ascontiguousarray(states, dtype=np.int64)
    cap = spec.nU << spec.D
    out = []
    step = max(1, (256 << 20) // (8 * cap))                 # 256 MB of offsets at a time
    for i in range(0, len(st_), step):
        part = st_[i:i + step]
        rec = np.empty(len(part) * cap, dtype=np.int64)
        st = l.orc_backup_states_touch(C.byref(p), part.ctypes.data, len(part), rec.ctypes.data, cap, nthreads or l.orc_max_threads())
        if st != 0:
            raise RuntimeError("orc_backup_states_touch status %d" % st)
        out.append(np.unique(rec))
    return np.unique(np.concatenate(out)) if out else np.empty(0, dtype=np.int64)


def backup_states_sparse(abi, spec, keys, vals, states, nthreads=0):
    """Backup of the listed states from J_next sampled at `keys` (sorted offsets, backup_states_touch) = `vals`:
    the checker for a J_next that lives on the device only (C3).  -> (J[k], idx[k])."""
    l = lib(abi)
    p, keep = spec.to_c()
    k = np.ascontiguousarray(keys, dtype=np.int64)
    v = np.ascontiguousarray(vals, dtype=np.float32)
    assert k.size == v.size and np.all(k[1:] > k[:-1])
    st_ = np.ascontiguousarray(states, dtype=np.int64)
    J = np.empty(len(st_), dtype=np.float32)
    idx = np.empty(len(st_), dtype=np.int32)
    st = l.orc_backup_states_sparse(C.byref(p), k.ctypes.data, v.ctypes.data, k.size, st_.ctypes.data, len(st_),
                                    J.ctypes.data, idx.ctypes.data, nthreads or l.orc_max_threads())
    if st != 0:
        raise RuntimeError("orc_backup_states_sparse status %d" % st)
    return J, idx


def canon_eval(abi, kind, a, b=None):
    """The fixed polynomial atan2 (kind 'atan2': a = y, b = x) / asin (kind 'asin') of the quaternion model."""
    l = lib(abi)
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(a if b is None else b, dtype=np.float32)
    out = np.empty_like(a)
    l.orc_canon_eval({"atan2": 0, "asin": 1}[kind], a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def backup_stage(abi, spec, J_next, slab=None, nthreads=0, impl="scalar"):
    """One canonical-arithmetic backup on the CPU.  Returns (J_out flat, idx).  impl: "scalar" (the twin every test
    compares against) or "avx2" (its row-vectorised form, float32 arithmetic only: the faster CPU baseline)."""
    l = lib(abi)
    p, keep = spec.to_c(slab)
    Jn = np.ascontiguousarray(np.asarray(J_next, dtype=spec.j_dtype).reshape(-1, order="F"))
    Jo = Jn.copy()
    if slab is None:
        n_owned = spec.nS
    else:
        n_owned = spec.nS // spec.n[-1] * (slab[1] - slab[0])
    idx = np.empty(n_owned, dtype=np.int32)
    nt = nthreads or l.orc_max_threads()
    fn = l.orc_backup_stage_avx2 if impl == "avx2" else l.orc_backup_stage
    st = fn(C.byref(p), Jn.ctypes.data, Jo.ctypes.data, idx.ctypes.data, nt)
    if st not in (0,):
        raise RuntimeError("orc_backup_stage status %d" % st)
    return Jo, idx


def sweep(abi, spec, n_stages, terminal=None, keep_J=False, keep_idx=False, monitor_period=0, monitor_tol=0.0,
          nthreads=0, monitor_single=False):
    """Labels come back as int32 whatever spec.idx_dtype says (compare with np.array_equal: it widens)."""
    l = lib(abi)
    p, keep = spec.to_c()
    nS, dt = spec.nS, spec.j_dtype
    o = abi.hjb_solve_opts()
    o.n_stages, o.monitor_period, o.monitor_tol = int(n_stages), int(monitor_period), float(monitor_tol)
    o.monitor_single = 1 if monitor_single else 0
    if terminal is not None:
        t = np.ascontiguousarray(np.asarray(terminal, dtype=dt).reshape(-1, order="F"))
        keep.append(t)
        o.terminal = t.ctypes.data
    J = np.empty(nS, dtype=dt)
    idx = np.empty(nS, dtype=np.int32)
    o.J_final, o.idx_final = J.ctypes.data, idx.ctypes.data
    Js = Is = None
    if keep_J:
        Js = np.zeros((nS, n_stages), dtype=dt, order="F")
        o.J_stages = Js.ctypes.data
    if keep_idx:
        Is = np.zeros((nS, n_stages), dtype=np.int32, order="F")
        o.idx_stages = Is.ctypes.data
    res = abi.hjb_result()
    nt = nthreads or l.orc_max_threads()
    st = l.orc_sweep(C.byref(p), C.byref(o), C.byref(res), nt)
    if st != 0:
        raise RuntimeError("orc_sweep status %d" % st)
    return {"J": J, "idx": idx, "J_stages": Js, "idx_stages": Is, "stages_done": res.stages_done,
            "stopped_early": bool(res.stopped_early), "last_e": res.last_e, "last_e2": res.last_e2}


def lookup(abi, knots, values, points, method="nearest"):
    """Canonical batched gridded lookup on the CPU (checker for hjb_policy_lookup)."""
    l = lib(abi)
    values = np.asarray(values)
    dt = values.dtype if values.dtype in (np.float32, np.float64) else np.dtype(np.float64)
    D = len(knots)
    V = np.ascontiguousarray(np.asarray(values, dtype=dt).reshape(-1, order="F"))
    Q = np.ascontiguousarray(np.asarray(points, dtype=dt).reshape(-1, D))
    ks = [np.ascontiguousarray(k, dtype=np.float64) for k in knots]
    n = (C.c_int32 * D)(*[len(k) for k in ks])
    kp = (C.POINTER(C.c_double) * D)(*[k.ctypes.data_as(C.POINTER(C.c_double)) for k in ks])
    out = np.empty(Q.shape[0], dtype=dt)
    st = l.orc_lookup(0 if dt == np.float32 else 1, D, n, kp, V.ctypes.data, Q.shape[0], Q.ctypes.data,
                      {"nearest": 0, "linear": 1}[method], out.ctypes.data)
    if st != 0:
        raise RuntimeError("orc_lookup status %d" % st)
    return out
