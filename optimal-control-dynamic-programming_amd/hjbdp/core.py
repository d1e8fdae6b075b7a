"""Thin host binding of libhjbdp (include/hjbdp.h) - the only way the host
classes reach the GPU.  There is NO CPU fallback: if the shared library is
missing, or no HIP device is visible, every compute call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import _abi
from .problem import ProblemSpec

_LIB = None
LIB_PATH = Path(__file__).resolve().parent / "libhjbdp.so"


class HjbError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("libhjbdp: %s (status %d)" % (text, status))
        self.status = status


def _share_torch_hip_runtime():
    """If PyTorch-ROCm is installed, map ITS bundled libamdhip64 (same soname,
    libamdhip64.so.7) before libhjbdp so the process has one HIP runtime: torch
    tensors, torch streams and RCCL then share a context with our kernels.  With
    ROCm's copy loaded first, a later `import torch` finds no GPU.  No torch
    installed -> the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
        if cand.exists():
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def load_library(path=None):
    """Load libhjbdp.so (built in-tree by __graft_entry__.build()).  Loading needs
    no GPU; raises if the file is missing - the product never falls back."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = Path(path) if path else Path(os.environ.get("HJBDP_LIB", LIB_PATH))
    if not p.exists():
        raise FileNotFoundError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). libhjbdp has no CPU fallback." % p)
    _share_torch_hip_runtime()
    lib = _abi.bind(C.CDLL(str(p)))
    if path is None:
        _LIB = lib
    return lib


def device_count():
    return int(load_library().hjb_device_count())


def _check(lib, handle, st):
    if st != _abi.HJB_OK:
        msg = lib.hjb_last_error(handle)
        raise HjbError(st, (msg or b"").decode() or lib.hjb_status_string(st).decode())


def policy_lookup(knots, values, points, method="nearest", device=0):
    """Batched griddedInterpolant(knots, values, method) at `points` [nq, D] on the GPU
    (hjb_policy_lookup).  values.dtype (float32/float64) is the arithmetic type."""
    lib = load_library()
    values = np.asarray(values)
    dt = values.dtype if values.dtype in (np.float32, np.float64) else np.dtype(np.float64)
    D = len(knots)
    V = np.ascontiguousarray(np.asarray(values, dtype=dt).reshape(-1, order="F"))
    Q = np.ascontiguousarray(np.asarray(points, dtype=dt).reshape(-1, D))
    nq = Q.shape[0]
    ks = [np.ascontiguousarray(k, dtype=np.float64) for k in knots]
    n = (C.c_int32 * D)(*[len(k) for k in ks])
    kp = (C.POINTER(C.c_double) * D)(*[k.ctypes.data_as(C.POINTER(C.c_double)) for k in ks])
    out = np.empty(nq, dtype=dt)
    meth = {"nearest": _abi.HJB_LOOKUP_NEAREST, "linear": _abi.HJB_LOOKUP_LINEAR}[method]
    st = lib.hjb_policy_lookup(int(device), _abi.HJB_F32 if dt == np.float32 else _abi.HJB_F64, D, n, kp,
                               V.ctypes.data, nq, Q.ctypes.data, meth, out.ctypes.data)
    _check(lib, None, st)
    return out


class DeviceBuffer:
    """A device allocation owned through the library (hjb_device_malloc): what a host without a HIP binding of its
    own hands to Backup.backup_stage_device.  `ptr` is an ordinary HIP device pointer."""

    def __init__(self, nbytes, device=0):
        self.lib = load_library()
        self.device, self.nbytes = int(device), int(nbytes)
        p = C.c_void_p()
        _check(self.lib, None, self.lib.hjb_device_malloc(self.device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def __int__(self):
        return self.ptr

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        _check(self.lib, None, self.lib.hjb_device_copy(self.device, self.ptr, a.ctypes.data, a.nbytes, _abi.HJB_COPY_H2D))

    def download(self, dtype, count=None):
        dt = np.dtype(dtype)
        out = np.empty(self.nbytes // dt.itemsize if count is None else int(count), dtype=dt)
        _check(self.lib, None, self.lib.hjb_device_copy(self.device, out.ctypes.data, self.ptr, out.nbytes, _abi.HJB_COPY_D2H))
        return out

    def gather(self, dtype, sel):
        """out[i] = buffer[sel[i]] (elements of `dtype`): sample a device-resident array."""
        dt = np.dtype(dtype)
        s = np.ascontiguousarray(sel, dtype=np.int64)
        out = np.empty(s.size, dtype=dt)
        _check(self.lib, None, self.lib.hjb_device_gather(self.device, self.ptr, dt.itemsize,
                                                          s.ctypes.data_as(C.POINTER(C.c_int64)), s.size, out.ctypes.data))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.hjb_device_free(self.device, self.ptr)
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_mem_info(device=0):
    lib = load_library()
    f, t = C.c_int64(), C.c_int64()
    _check(lib, None, lib.hjb_device_mem_info(int(device), C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


class Backup:
    """A problem resident on one GPU.  Mirrors the C handle one to one."""

    def __init__(self, spec: ProblemSpec, device=0, slab=None, variant=None):
        self.lib = load_library()
        self.spec = spec
        self._cprob, self._keep = spec.to_c(slab)
        self._h = C.c_void_p()
        st = self.lib.hjb_create(C.byref(self._cprob), int(device), C.byref(self._h))
        if st != _abi.HJB_OK:
            self._h = C.c_void_p()
            _check(self.lib, None, st)
        self.device = int(device)
        if variant is not None:
            self.set_option("variant", variant)

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.hjb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- queries ----------------------------------------------------------
    def info(self):
        inf = _abi.hjb_info()
        _check(self.lib, self._h, self.lib.hjb_get_info(self._h, C.byref(inf)))
        return {k: getattr(inf, k) for k, _ in inf._fields_}

    def set_option(self, key, value):
        _check(self.lib, self._h, self.lib.hjb_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64()
        _check(self.lib, self._h, self.lib.hjb_get_option(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    # -- one stage, host buffers -------------------------------------------
    def backup_stage(self, J_next):
        """[J_k, idx] = min(g + F(x_next), [], ctrl): J_next/J_k in the (haloed)
        column-major J layout, idx int32 labels of the owned states."""
        inf = self.info()
        dt = self.spec.j_dtype
        Jn = np.ascontiguousarray(np.asarray(J_next, dtype=dt).reshape(-1, order="F"))
        if Jn.size != inf["j_elems"]:
            raise ValueError("J_next has %d elements, the handle's J layout has %d" % (Jn.size, inf["j_elems"]))
        Jo = np.empty_like(Jn)
        idx = np.empty(inf["n_states"], dtype=self.spec.idx_np_dtype)
        st = self.lib.hjb_backup_stage(self._h, Jn.ctypes.data, Jo.ctypes.data, idx.ctypes.data)
        _check(self.lib, self._h, st)
        return Jo, idx

    # -- one stage, device buffers (torch tensors or raw pointers) -----------
    def backup_stage_device(self, dJ_next, dJ_out, d_idx=None, stream=0):
        def ptr(x):
            if x is None:
                return None
            return int(x.data_ptr()) if hasattr(x, "data_ptr") else int(x)
        st = self.lib.hjb_backup_stage_device(self._h, ptr(dJ_next), ptr(dJ_out), ptr(d_idx), int(stream) or None)
        _check(self.lib, self._h, st)

    def fill_separable(self, vecs, dJ, stream=0):
        """dJ[s] = ((vecs[0][i0] + vecs[1][i1]) + ...) over the whole grid (hjb_device_fill_separable)."""
        vs = [np.ascontiguousarray(v, dtype=self.spec.dtype) for v in vecs]
        ptrs = (C.c_void_p * len(vs))(*[v.ctypes.data for v in vs])
        dp = int(dJ.data_ptr()) if hasattr(dJ, "data_ptr") else int(dJ)
        _check(self.lib, self._h, self.lib.hjb_device_fill_separable(self._h, ptrs, dp, int(stream) or None))

    def check_device_status(self, stream=0):
        _check(self.lib, self._h, self.lib.hjb_check_device_status(self._h, int(stream) or None))

    # -- the whole sweep -----------------------------------------------------
    def solve(self, n_stages, terminal=None, keep_J=False, keep_idx=False, monitor_period=0, monitor_tol=0.0,
              progress=None, progress_every_stage=False, probe=None, monitor_single=False):
        """Backward sweep of n_stages backups.  Returns a dict with J (final), idx
        (final), optional J_stages/idx_stages [nS, n_stages] with reference stage
        k_s at column k_s-1, stages_done, stopped_early, sweep_ms.
        probe = {"lo": [..D], "hi": [..D], "control": [..C], "want": ("g", "x_next", "j_interp")} (0-based,
        half-open) asks for the reference's debug taps (Dynamic_Solver.m:212-219) of every stage: out["probe"]
        holds g [block..., n_stages], x_next [block..., D, n_stages], j_interp [block..., n_stages]."""
        nS, dt = self.spec.nS, self.spec.j_dtype
        o = _abi.hjb_solve_opts()
        o.n_stages = int(n_stages)
        o.monitor_period = int(monitor_period)
        o.monitor_tol = float(monitor_tol)
        keep = []
        if terminal is not None:
            t = np.ascontiguousarray(np.asarray(terminal, dtype=dt).reshape(-1, order="F"))
            if t.size != nS:
                raise ValueError("terminal cost must have nS elements")
            keep.append(t)
            o.terminal = t.ctypes.data
        J = np.empty(nS, dtype=dt)
        idx = np.empty(nS, dtype=self.spec.idx_np_dtype)
        o.J_final, o.idx_final = J.ctypes.data, idx.ctypes.data
        Js = Is = None
        if keep_J:
            Js = np.zeros((nS, n_stages), dtype=dt, order="F")
            o.J_stages = Js.ctypes.data
        if keep_idx:
            Is = np.zeros((nS, n_stages), dtype=self.spec.idx_np_dtype, order="F")
            o.idx_stages = Is.ctypes.data
        if progress is not None:
            cb = _abi.hjb_progress_fn(lambda user, k_s, e, e2, sec: progress(k_s, e, e2, sec))
            keep.append(cb)
            o.progress = cb
        o.progress_every_stage = 1 if progress_every_stage else 0
        o.monitor_single = 1 if monitor_single else 0
        pout = None
        if probe is not None:
            pout, pb = self._make_probe(probe, n_stages)
            keep.append(pb)
            o.probe = C.pointer(pb)
        res = _abi.hjb_result()
        st = self.lib.hjb_solve(self._h, C.byref(o), C.byref(res))
        _check(self.lib, self._h, st)
        return {"J": J, "idx": idx, "J_stages": Js, "idx_stages": Is, "stages_done": res.stages_done,
                "stopped_early": bool(res.stopped_early), "sweep_ms": res.sweep_ms, "last_e": res.last_e,
                "last_e2": res.last_e2, "probe": pout}

    def _make_probe(self, probe, n_planes):
        D, Cc = self.spec.D, self.spec.C
        pb = _abi.hjb_probe()
        ext = []
        for a in range(D):
            pb.lo[a], pb.hi[a] = int(probe["lo"][a]), int(probe["hi"][a])
            ext.append(pb.hi[a] - pb.lo[a])
        for c in range(Cc):
            pb.control[c] = int(probe["control"][c])
        want = probe.get("want", ("g", "x_next", "j_interp"))
        dt = self.spec.dtype
        tail = (n_planes,) if n_planes else ()
        out = {}
        if "g" in want:
            out["g"] = np.zeros(tuple(ext) + tail, dtype=dt, order="F")
            pb.g = out["g"].ctypes.data
        if "x_next" in want:
            out["x_next"] = np.zeros(tuple(ext) + (D,) + tail, dtype=dt, order="F")
            pb.x_next = out["x_next"].ctypes.data
        if "j_interp" in want:
            out["j_interp"] = np.zeros(tuple(ext) + tail, dtype=dt, order="F")
            pb.j_interp = out["j_interp"].ctypes.data
        return out, pb

    def probe_stage(self, probe, J_next=None):
        """One stage of the debug taps from a host J_next (hjb_probe_stage)."""
        out, pb = self._make_probe(probe, 0)
        Jn = None
        if J_next is not None:
            Jn = np.ascontiguousarray(np.asarray(J_next, dtype=self.spec.j_dtype).reshape(-1, order="F"))
        elif "j_interp" in out:
            raise ValueError("j_interp needs J_next")
        st = self.lib.hjb_probe_stage(self._h, Jn.ctypes.data if Jn is not None else None, C.byref(pb))
        _check(self.lib, self._h, st)
        return out


class MultiBackup:
    """A problem partitioned along its last state axis over several devices of THIS process (hjb_create_multi /
    hjb_solve_multi): per stage the halo planes travel device to device while the interior planes are computed.
    `devices` may repeat a device (several slabs on one GPU: how the path is tested on a 1-GPU box)."""

    def __init__(self, spec: ProblemSpec, devices):
        self.lib = load_library()
        self.spec = spec
        self._cprob, self._keep = spec.to_c()
        self._m = C.c_void_p()
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        st = self.lib.hjb_create_multi(C.byref(self._cprob), len(devices), devs, C.byref(self._m))
        if st != _abi.HJB_OK:
            self._m = C.c_void_p()
            msg = self.lib.hjb_multi_last_error(None)
            raise HjbError(st, (msg or b"").decode() or self.lib.hjb_status_string(st).decode())
        self.n_slabs = len(devices)

    def _check(self, st):
        if st != _abi.HJB_OK:
            msg = self.lib.hjb_multi_last_error(self._m)
            raise HjbError(st, (msg or b"").decode() or self.lib.hjb_status_string(st).decode())

    def close(self):
        if getattr(self, "_m", None) and self._m.value:
            self.lib.hjb_destroy_multi(self._m)
            self._m = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def slab_info(self, i):
        v = [C.c_int32() for _ in range(6)]
        self._check(self.lib.hjb_multi_slab_info(self._m, int(i), *[C.byref(x) for x in v]))
        return dict(zip(("begin", "end", "halo_lo", "halo_hi", "split", "kernel_variant"), (x.value for x in v)))

    def set_option(self, key, value):
        self._check(self.lib.hjb_multi_set_option(self._m, key.encode(), int(value)))

    def solve(self, n_stages, terminal=None, keep_J=False, keep_idx=False, monitor_period=0, monitor_tol=0.0, progress=None,
              progress_every_stage=False):
        nS, dt = self.spec.nS, self.spec.j_dtype
        o = _abi.hjb_solve_opts()
        o.n_stages, o.monitor_period, o.monitor_tol = int(n_stages), int(monitor_period), float(monitor_tol)
        keep = []
        if terminal is not None:
            t = np.ascontiguousarray(np.asarray(terminal, dtype=dt).reshape(-1, order="F"))
            if t.size != nS:
                raise ValueError("terminal cost must have nS elements")
            keep.append(t)
            o.terminal = t.ctypes.data
        J = np.empty(nS, dtype=dt)
        idx = np.empty(nS, dtype=self.spec.idx_np_dtype)
        o.J_final, o.idx_final = J.ctypes.data, idx.ctypes.data
        Js = Is = None
        if keep_J:
            Js = np.zeros((nS, n_stages), dtype=dt, order="F")
            o.J_stages = Js.ctypes.data
        if keep_idx:
            Is = np.zeros((nS, n_stages), dtype=self.spec.idx_np_dtype, order="F")
            o.idx_stages = Is.ctypes.data
        if progress is not None:
            cb = _abi.hjb_progress_fn(lambda user, k_s, e, e2, sec: progress(k_s, e, e2, sec))
            keep.append(cb)
            o.progress = cb
        o.progress_every_stage = 1 if progress_every_stage else 0
        res = _abi.hjb_result()
        self._check(self.lib.hjb_solve_multi(self._m, C.byref(o), C.byref(res)))
        return {"J": J, "idx": idx, "J_stages": Js, "idx_stages": Is, "stages_done": res.stages_done, "stopped_early": bool(res.stopped_early),
                "sweep_ms": res.sweep_ms, "last_e": res.last_e, "last_e2": res.last_e2}


class RankSlab:
    """One rank's share of a sweep partitioned over `world` processes, one per GPU (hjb_rank_create / hjb_rank_stage):
    the library builds the slab handle and - with overlap and an interior - the interior and strip handles, and enqueues
    a whole stage (interior on the compute stream, strips behind the halos on streams of their own) in one call."""

    def __init__(self, spec: ProblemSpec, device, rank, world, overlap=True):
        self.lib = load_library()
        self.spec = spec
        self._cprob, self._keep = spec.to_c()
        self._r = C.c_void_p()
        st = self.lib.hjb_rank_create(C.byref(self._cprob), int(device), int(rank), int(world), 1 if overlap else 0, C.byref(self._r))
        if st != _abi.HJB_OK:
            self._r = C.c_void_p()
            msg = self.lib.hjb_rank_last_error(None)
            raise HjbError(st, (msg or b"").decode() or self.lib.hjb_status_string(st).decode())
        self.refresh()

    def refresh(self):
        """Re-read what the library reports for this rank (the kernel variant changes with set_option)."""
        v = (C.c_int32 * 10)()
        self._check(self.lib.hjb_rank_info(self._r, v))
        (self.begin, self.end, self.halo_lo, self.halo_hi, self.split, self.kernel_variant, self.need_lo, self.need_hi,
         self.idx_bytes, self.n_planes) = (int(x) for x in v)

    def _check(self, st):
        if st != _abi.HJB_OK:
            msg = self.lib.hjb_rank_last_error(self._r)
            raise HjbError(st, (msg or b"").decode() or self.lib.hjb_status_string(st).decode())

    def stage(self, dJ_in, dJ_out, d_idx, compute_stream=0, halo_stream=0):
        def ptr(x):
            if x is None:
                return None
            return int(x.data_ptr()) if hasattr(x, "data_ptr") else int(x)
        self._check(self.lib.hjb_rank_stage(self._r, ptr(dJ_in), ptr(dJ_out), ptr(d_idx), int(compute_stream) or None,
                                            int(halo_stream) or None))

    def fill_separable(self, vecs, dJ, stream=0):
        """dJ (this rank's haloed buffer) = ((vecs[0][i0] + vecs[1][i1]) + ...) on the rank's planes, halos included; vecs are the
        GLOBAL vectors (hjb_rank_fill_separable)."""
        vs = [np.ascontiguousarray(v, dtype=self.spec.dtype) for v in vecs]
        ptrs = (C.c_void_p * len(vs))(*[v.ctypes.data for v in vs])
        dp = int(dJ.data_ptr()) if hasattr(dJ, "data_ptr") else int(dJ)
        self._check(self.lib.hjb_rank_fill_separable(self._r, ptrs, dp, int(stream) or None))

    def stage_post(self, dJ_in, dJ_out, d_idx, compute_stream=0, halo_stream=0):
        """The stage with the boundary strips first (hjb_rank_stage_post): their halos are already in dJ_in."""
        def ptr(x):
            if x is None:
                return None
            return int(x.data_ptr()) if hasattr(x, "data_ptr") else int(x)
        self._check(self.lib.hjb_rank_stage_post(self._r, ptr(dJ_in), ptr(dJ_out), ptr(d_idx), int(compute_stream) or None,
                                                 int(halo_stream) or None))

    def wait_strips(self, stream):
        """`stream` waits for the last stage's boundary strips; -> True when they cover every plane a neighbour needs."""
        cov = C.c_int32(0)
        self._check(self.lib.hjb_rank_wait_strips(self._r, int(stream) or None, C.byref(cov)))
        return bool(cov.value)

    def set_option(self, key, value):
        self._check(self.lib.hjb_rank_set_option(self._r, key.encode(), int(value)))
        self.refresh()

    def get_option(self, key):
        v = C.c_int64()
        self._check(self.lib.hjb_rank_get_option(self._r, key.encode(), C.byref(v)))
        return int(v.value)

    def check_device_status(self, stream=0):
        self._check(self.lib.hjb_rank_check_status(self._r, int(stream) or None))

    def close(self):
        if getattr(self, "_r", None) and self._r.value:
            self.lib.hjb_rank_destroy(self._r)
            self._r = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def suggest_axis_order(spec):
    """The labelling of the state axes under which the library runs its fastest stage kernel on `spec`
    (hjb_problem_suggest_order, the call a MATLAB host makes through matlab/hjbdp_solve.m): a tuple for
    `permute_state_axes`, or None when there is nothing to gain.  Needs the library, not a GPU."""
    lib = load_library()
    if spec.model is not None:
        return None
    b = C.c_void_p()
    n = (C.c_int32 * spec.D)(*spec.n)
    m = (C.c_int32 * spec.C)(*spec.m)
    dt = _abi.HJB_F64 if spec.dtype == np.float64 else _abi.HJB_F32

    def ok(st):
        if st != _abi.HJB_OK:
            msg = lib.hjb_problem_last_error(b)
            raise HjbError(st, (msg or b"").decode())
    ok(lib.hjb_problem_new(spec.D, spec.C, n, m, dt, spec.index_base, C.byref(b)))
    try:
        if spec.table_dtype is not None:
            ok(lib.hjb_problem_set_types(b, _abi.HJB_IDX_I32, _abi.HJB_TAB_F64))
        for a in range(spec.D):
            k = np.ascontiguousarray(spec.knots[a], dtype=np.float64)
            ok(lib.hjb_problem_set_knots(b, a, k.ctypes.data_as(C.POINTER(C.c_double)), k.size))
            for t in spec.next_terms[a]:
                v = np.ascontiguousarray(np.asarray(t.data, dtype=spec.table_dtype or spec.dtype).reshape(-1, order="F"))
                ok(lib.hjb_problem_add_next_term(b, a, sum(1 << d for d in t.dims), v.ctypes.data, v.size))
        order = (C.c_int32 * spec.D)()
        found = C.c_int32(0)
        ok(lib.hjb_problem_suggest_order(b, order, C.byref(found)))
        return tuple(order) if found.value else None
    finally:
        lib.hjb_problem_free(b)


def solve_batch(specs, n_stages, device=0, monitor_period=0, monitor_tol=0.0, progress=None, monitor_single=False, cs_split=None):
    """Independent sweeps side by side with as few launch chains as the library can make of them (the four channels of
    Solver_pos_att.simplified_run, pos-att/Solver_pos_att.m:197-242): problems that run on the column-sweep kernel with the same group
    axis, or on the table kernel's 32-bit form with one (dtype, D), share ONE launch per stage (hjb_solve_batch); each such group, and
    every problem that is alone in its shape, gets a host thread and a stream of its own (the device runs two launch chains at full rate: three channels + one is two chains).  Every
    problem keeps its own monitor sums and stop decision; results equal Backup.solve's bit for bit.
    -> (outs, wall_ms, variants, group sizes)"""
    import time
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    lib = load_library()
    n = len(specs)
    kw = dict(monitor_period=monitor_period, monitor_tol=monitor_tol, progress=progress, monitor_single=monitor_single)
    with ThreadPoolExecutor(max_workers=max(1, n)) as ex:
        bks = list(ex.map(lambda s: Backup(s, device=device), specs))
    t_made = time.perf_counter()
    t_run = t_made
    try:
        variants = [bk.info()["kernel_variant"] for bk in bks]
        groups = {}
        for i, bk in enumerate(bks):
            if variants[i] == 7:
                key = ("colsweep", bk.get_option("cs_group_axis"), bk.spec.cost_dtype is not None)
            elif variants[i] == 5 and bk.spec.D <= 4:
                key = ("tabled", np.dtype(bk.spec.j_dtype).str, bk.spec.D)
            else:
                key = ("alone", i)
            groups.setdefault(key, []).append(i)
        # The table kernel is one state per thread: a batch pays off while ALL its states are resident at once (a launch-bound stage);
        # beyond one round of the wave slots (256 CUs x 32 waves x 64 lanes) the launches run round after round and three chains on
        # three streams overlap better than one launch (Solver_attitude.simplified_run's 3 x 3e5 states: 130 ms batched, 82 ms as
        # chains - profiles/r06_batch_attitude.log)
        for key in [k for k in groups if k[0] == "tabled"]:
            if len(groups[key]) > 1 and sum(specs[i].nS for i in groups[key]) > 256 * 32 * 64:
                for i in groups.pop(key):
                    groups[("alone", i)] = [i]
        outs = [None] * n
        if cs_split is None:
            # Parts per column of the column-sweep problems.  A handle alone picks as many parts as fill the device (it is one wave's
            # chain of round trips), and every part primes its rows again: up to twice the work per column.  These problems are in
            # flight TOGETHER, so the parts are what lets all their columns fill about one round of the wave slots - the library does
            # the same inside a batch for the columns it sees; here every chain of the call is counted (profiles/r06_batch_split.log)
            cs = [i for i in range(n) if variants[i] == 7]
            if len(cs) > 1:
                cols = sum(-(-specs[i].n[0] // 60) * specs[i].n[2] * specs[i].n[3] for i in cs)
                slots = (5 if any(specs[i].cost_dtype is not None for i in cs) else 6) * 4 * 256          # waves per SIMD of the form that runs (86 / 80 registers) x SIMDs
                cs_split = max(1, slots // max(cols, 1))
                cs_split = cs_split if all(cs_split < bks[i].get_option("cs_split") for i in cs) else 0

        def run(members):
            if cs_split:
                for i in members:
                    if variants[i] == 7:
                        bks[i].set_option("cs_split", int(cs_split))
            if len(members) > 1:
                m = len(members)
                keep, bufs, ress = [], [], []
                hs = (C.c_void_p * m)(*[bks[i]._h for i in members])
                optp = (C.POINTER(_abi.hjb_solve_opts) * m)()
                resp = (C.POINTER(_abi.hjb_result) * m)()
                for k, i in enumerate(members):
                    s = specs[i]
                    o = _abi.hjb_solve_opts()
                    o.n_stages, o.monitor_period, o.monitor_tol = int(n_stages), int(monitor_period), float(monitor_tol)
                    J = np.empty(s.nS, dtype=s.j_dtype)
                    idx = np.empty(s.nS, dtype=s.idx_np_dtype)
                    o.J_final, o.idx_final = J.ctypes.data, idx.ctypes.data
                    o.monitor_single = 1 if monitor_single else 0
                    if progress is not None:
                        cb = _abi.hjb_progress_fn(lambda user, k_s, e, e2, sec: progress(k_s, e, e2, sec))
                        keep.append(cb)
                        o.progress = cb
                    r = _abi.hjb_result()
                    keep += [o, r]
                    optp[k] = C.pointer(o)
                    resp[k] = C.pointer(r)
                    bufs.append((J, idx))
                    ress.append(r)
                st = lib.hjb_solve_batch(m, hs, optp, resp)
                if st == _abi.HJB_OK:
                    for i, (J, idx), r in zip(members, bufs, ress):
                        outs[i] = {"J": J, "idx": idx, "J_stages": None, "idx_stages": None, "stages_done": r.stages_done,
                                   "stopped_early": bool(r.stopped_early), "sweep_ms": r.sweep_ms, "last_e": r.last_e, "last_e2": r.last_e2,
                                   "probe": None}
                    return [m]
                if st != _abi.HJB_E_UNSUPPORTED:
                    _check(lib, None, st)
            def one(i):                     # alone in its shape (or a group the library did not take): the plain sweep
                outs[i] = bks[i].solve(n_stages, **kw)
            if len(members) == 1:
                one(members[0])
            else:
                with ThreadPoolExecutor(max_workers=len(members)) as ex2:
                    list(ex2.map(one, members))
            return [1] * len(members)
        with ThreadPoolExecutor(max_workers=max(1, len(groups))) as ex:
            sizes = [m for ms in ex.map(run, groups.values()) for m in ms]
        t_run = time.perf_counter()
    finally:
        for bk in bks:
            bk.close()
    t_end = time.perf_counter()
    solve_batch.last_phases_ms = {"create": (t_made - t0) * 1e3, "sweep": (t_run - t_made) * 1e3, "close": (t_end - t_run) * 1e3}
    return outs, (t_end - t0) * 1e3, variants, sizes


def solve_many(specs, n_stages, device=0, **solve_kw):
    """Independent sweeps (the three axis channels of Solver_position / Solver_attitude.simplified_run, the four
    runs of Solver_pos_att.simplified_run) in flight together: one handle and one host thread per sweep, each on
    its handle's own HIP stream.  A channel's stage kernel fills a few percent of the chip and the sweep is a
    chain of thousands of dependent launches, so running channels side by side costs nothing and divides the
    wall time.  `solve_kw` values may be lists (one entry per spec).  Returns (outs, wall_ms, variants)."""
    import time
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        kw = {k: (v[i] if isinstance(v, (list, tuple)) else v) for k, v in solve_kw.items()}
        with Backup(specs[i], device=device) as bk:
            out = bk.solve(n_stages, **kw)
            return out, bk.info()["kernel_variant"]
    t0 = time.perf_counter()
    if len(specs) == 1:
        res = [one(0)]
    else:
        with ThreadPoolExecutor(max_workers=len(specs)) as ex:
            res = list(ex.map(one, range(len(specs))))
    wall_ms = (time.perf_counter() - t0) * 1e3
    return [r[0] for r in res], wall_ms, [r[1] for r in res]

