"""The MATLAB shims' call sequences, replayed through ctypes (test infrastructure; the build image has no MATLAB).

`hjbdp_solve` below is matlab/hjbdp_solve.m statement for statement: the same flat entry points (include/hjbdp_matlab.h) in
the same order with the same argument values, `prob` being the struct the .m files build (1-based `dims`, column-major
`data`).  The `*_prob` functions restate the problem set-up of the four solver shims -
matlab/Solver_position_hjbdp_simplified_run.m, Solver_attitude_hjbdp_simplified_run.m, Solver_attitude_hjbdp_run.m,
Solver_pos_att_hjbdp_channel.m - line for line, taking the reference objects' property values from the Python mirrors'
constructors (hjbdp/solver_*.py restate the reference constructors).  tests/test_gpu_flat_api.py runs them on the GPU and
compares with the Python mirrors (which are oracle-checked); tests/test_abi.py checks every calllib in the .m files against
the header and that this file calls exactly the functions hjbdp_solve.m calls."""
import ctypes as C
import time

import numpy as np

f32 = np.float32


def T(dims, data):
    return {"dims": [dims] if np.isscalar(dims) else list(dims), "data": np.asarray(data)}


def hjbdp_solve(lib, prob, n_stages, keep_stages=False, monitor_period=0, monitor_tol=0.0, devices=0, fast_axes=True,
                double_tables=False, monitor_single=False, labels="int32", double_cost=False, on_stage=None):
    D, Cn = len(prob["knots"]), len(prob["m"])
    cls, dt = (np.float32, 0) if prob["single"] else (np.float64, 1)
    n = [len(k) for k in prob["knots"]]
    b = C.c_void_p()

    def check(st, obj, kind):
        if st == 0:
            return
        msg = {"builder": lib.hjb_problem_last_error, "multi": lib.hjb_multi_last_error}.get(kind, lib.hjb_last_error)(obj)
        raise RuntimeError("%s (%s)" % ((msg or b"").decode(), lib.hjb_status_string(st).decode()))

    check(lib.hjb_problem_new(D, Cn, (C.c_int32 * D)(*n), (C.c_int32 * Cn)(*prob["m"]), dt, 1, C.byref(b)), None, "builder")
    try:
        ncls = cls
        top = int(np.prod(prob["m"]))
        idt = {"int32": 0, "uint8": 1, "uint16": 2, "auto": 3}[labels]
        icls = np.int32
        if idt == 1 or (idt == 3 and top <= 255):
            icls = np.uint8
        elif idt == 2 or (idt == 3 and top <= 65535):
            icls = np.uint16
        if double_tables and not prob["single"]:
            raise ValueError("double_tables is for prob.single = true")
        if double_tables or idt != 0:
            check(lib.hjb_problem_set_types(b, idt, 1 if double_tables else 0), b, "builder")
        if double_tables:
            ncls = np.float64
        ccls = cls
        if double_cost:
            if not prob["single"]:
                raise ValueError("double_cost is for prob.single = true")
            check(lib.hjb_problem_set_cost_type(b, 1), b, "builder")
            ccls = np.float64
        keep = []
        if prob.get("model"):
            tb = [np.ascontiguousarray(np.asarray(t, dtype=f32).reshape(-1, order="F")) for t in prob["model"]["tables"]]
            keep += tb
            check(lib.hjb_problem_set_model(b, 1, float(prob["model"]["h"]), *[t.ctypes.data for t in tb]), b, "builder")

        def mask(dims):
            return int(sum(1 << (d - 1) for d in dims))

        for a in range(D):
            k = np.ascontiguousarray(prob["knots"][a], dtype=np.float64)
            check(lib.hjb_problem_set_knots(b, a, k.ctypes.data_as(C.POINTER(C.c_double)), n[a]), b, "builder")
            for t in prob["next_terms"][a]:
                v = np.ascontiguousarray(np.asarray(t["data"]).reshape(-1, order="F").astype(ncls))
                check(lib.hjb_problem_add_next_term(b, a, mask(t["dims"]), v.ctypes.data, v.size), b, "builder")
        for t in prob["cost_terms"]:
            v = np.ascontiguousarray(np.asarray(t["data"]).reshape(-1, order="F").astype(ccls))
            check(lib.hjb_problem_add_cost_term(b, mask(t["dims"]), v.ctypes.data, v.size), b, "builder")
        order = list(range(D))
        if fast_axes and D > 1:
            ord0, found = (C.c_int32 * D)(), C.c_int32(0)
            check(lib.hjb_problem_suggest_order(b, ord0, C.byref(found)), b, "builder")
            if found.value:
                check(lib.hjb_problem_permute_axes(b, ord0), b, "builder")
                order = list(ord0)
        nS = int(np.prod(n))
        term = None
        if prob.get("terminal") is not None:
            term = np.asarray(prob["terminal"], dtype=cls).reshape(n, order="F")
            term = np.ascontiguousarray(np.transpose(term, order).reshape(-1, order="F"))
        Jf, If = np.zeros(nS, dtype=cls), np.zeros(nS, dtype=icls)
        done, early, ms = C.c_int32(), C.c_int32(), C.c_double()
        h = C.c_void_p()
        Js = Is = None
        if np.isscalar(devices):
            check(lib.hjb_create_from(b, int(devices), C.byref(h)), b, "builder")
            try:
                if monitor_single:
                    check(lib.hjb_set_option(h, b"monitor_single", 1), h, "handle")
                if keep_stages:
                    Js = np.zeros(nS * n_stages, dtype=cls)
                    Is = np.zeros(nS * n_stages, dtype=icls)
                if on_stage is None:
                    check(lib.hjb_solve_flat(h, n_stages, monitor_period, float(monitor_tol), None if term is None else term.ctypes.data,
                                             Jf.ctypes.data, If.ctypes.data, None if Js is None else Js.ctypes.data,
                                             None if Is is None else Is.ctypes.data, C.byref(done), C.byref(early), C.byref(ms)), h, "handle")
                else:       # the caller's own stage loop on device buffers
                    if keep_stages or monitor_period > 0:
                        raise ValueError("on_stage runs without keep_stages and monitor")
                    dev = int(devices)
                    jbytes, ibytes = nS * (4 if prob["single"] else 8), nS * np.dtype(icls).itemsize
                    dJ, dI = [C.c_void_p(), C.c_void_p()], C.c_void_p()
                    try:
                        check(lib.hjb_device_malloc(dev, jbytes, C.byref(dJ[0])), h, "handle")
                        check(lib.hjb_device_malloc(dev, jbytes, C.byref(dJ[1])), h, "handle")
                        check(lib.hjb_device_malloc(dev, ibytes, C.byref(dI)), h, "handle")
                        if term is None:
                            term = np.zeros(nS, dtype=cls)
                        check(lib.hjb_device_copy(dev, dJ[0], term.ctypes.data, jbytes, 0), h, "handle")
                        cur, nd, t0 = 0, 0, time.time()
                        for k_s in range(n_stages, 0, -1):
                            check(lib.hjb_backup_stage_device(h, dJ[cur], dJ[1 - cur], dI, None), h, "handle")
                            cur, nd = 1 - cur, nd + 1
                            if on_stage(k_s):
                                break
                        check(lib.hjb_check_device_status(h, None), h, "handle")
                        ms.value, done.value, early.value = 1e3 * (time.time() - t0), nd, int(nd < n_stages)
                        check(lib.hjb_device_copy(dev, Jf.ctypes.data, dJ[cur], jbytes, 1), h, "handle")
                        check(lib.hjb_device_copy(dev, If.ctypes.data, dI, ibytes, 1), h, "handle")
                    finally:
                        for q in (dJ[0], dJ[1], dI):
                            if q:
                                lib.hjb_device_free(dev, q)
            finally:
                lib.hjb_destroy(h)
        else:
            if keep_stages:
                raise ValueError("keep_stages needs a single device")
            dv = (C.c_int32 * len(devices))(*devices)
            check(lib.hjb_create_multi_from(b, len(devices), dv, C.byref(h)), b, "builder")
            try:
                check(lib.hjb_solve_multi_flat(h, n_stages, monitor_period, float(monitor_tol), None if term is None else term.ctypes.data,
                                               Jf.ctypes.data, If.ctypes.data, C.byref(done), C.byref(early), C.byref(ms)), h, "multi")
            finally:
                lib.hjb_destroy_multi(h)
    finally:
        lib.hjb_problem_free(b)
    shape = [n[i] for i in order]
    inv = np.argsort(order)

    def back(v):        # ipermute(reshape(v, shape), order)
        return np.transpose(np.asarray(v).reshape(shape, order="F"), inv)

    out = {"J": back(Jf), "idx": back(If.astype(np.float64)), "axis_order": [o + 1 for o in order],
           "stages_done": done.value, "stopped_early": bool(early.value), "sweep_ms": ms.value}
    if keep_stages:
        out["J_stages"] = np.stack([back(Js[k * nS:(k + 1) * nS]).reshape(-1, order="F") for k in range(n_stages)], axis=1)
        out["idx_stages"] = np.stack([back(Is[k * nS:(k + 1) * nS].astype(np.float64)).reshape(-1, order="F")
                                      for k in range(n_stages)], axis=1)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the reference classes' helper methods the shims call (on vectors)

def _rk4_shift(k1, h):        # X + h*(k1 + 2*k2 + 2*k3 + k4)/6 with X = 0 and the k's fed back (Solver_position.m:157-167)
    k2 = k1 + k1 * h / 2
    k3 = k1 + k2 * h / 2
    k4 = k1 + k3 * h
    return np.zeros_like(k1) + h * (k1 + 2 * k2 + 2 * k3 + k4) / 6


def _rk4_const(k, h):         # every k equal (Solver_position.m:173-182, Solver_attitude.m:630-644)
    return np.zeros_like(k) + h * (k + 2 * k + 2 * k + k) / 6


def position_channel_prob(sp, ch):
    """matlab/Solver_position_hjbdp_simplified_run.m, one channel.  sp: hjbdp.Solver_position (constructor values)."""
    from hjbdp.matlab_compat import sym_linspace_position
    x = sym_linspace_position(sp.x_min, sp.x_max, sp.n_mesh_x)
    v = sym_linspace_position(sp.v_min, sp.v_max, sp.n_mesh_v)
    Qx, Qv, R = (sp.Qx1, sp.Qx2, sp.Qx3)[ch], (sp.Qv1, sp.Qv2, sp.Qv3)[ch], (sp.R1, sp.R2, sp.R3)[ch]
    U = np.asarray(sp.U_vector, dtype=np.float64)
    dx = _rk4_shift(v, sp.h)                          # RK4_x(obj, zeros(size(v)), v, obj.h)
    dv = _rk4_const(U / sp.Mass, sp.h)                # RK4_v(obj, zeros(size(U)), U, obj.h)
    return {"knots": [x, v], "m": [len(U)], "single": False,
            "next_terms": [[T(1, x), T(2, dx)], [T(2, v), T(3, dv)]],
            "cost_terms": [T(1, Qx * x ** 2), T(2, Qv * v ** 2), T(3, R * U ** 2)]}


def attitude_simplified_prob(sa, ch):
    """matlab/Solver_attitude_hjbdp_simplified_run.m, one channel (the reference's n_mesh_w is the mirror's
    n_mesh_w_simplified)."""
    from hjbdp.matlab_compat import deg2rad, linspace
    s_w = linspace(sa.w_min, sa.w_max, sa.n_mesh_w_simplified)
    lims = [(sa.yaw_min, sa.yaw_max), (sa.pitch_min, sa.pitch_max), (sa.roll_min, sa.roll_max)][ch]
    t = linspace(float(deg2rad(lims[0])), float(deg2rad(lims[1])), sa.n_mesh_t)
    Jc, Qw, Qt, R = (sa.J1, sa.J2, sa.J3)[ch], (sa.Q1, sa.Q2, sa.Q3)[ch], (sa.Qt1, sa.Qt2, sa.Qt3)[ch], (sa.R1, sa.R2, sa.R3)[ch]
    U = np.asarray(sa.U_vector, dtype=np.float64)
    dw = _rk4_const(U / Jc, sa.h)                     # RK4_w(obj, zeros(size(U)), U, Jc, obj.h)
    dtt = _rk4_shift(s_w, sa.h)                       # RK4_t(obj, zeros(size(s_w)), s_w, obj.h)
    return {"knots": [s_w, t], "m": [len(U)], "single": False,
            "next_terms": [[T(1, s_w), T(3, dw)], [T(2, t), T(1, dtt)]],
            "cost_terms": [T(1, Qw * s_w ** 2), T(2, Qt * t ** 2), T(3, R * U ** 2)]}


def attitude_run_prob(sa, on_the_fly=True):
    """matlab/Solver_attitude_hjbdp_run.m: axes (yaw, pitch, roll, w1, w2, w3), single typed operands."""
    from hjbdp.matlab_compat import deg2rad, linspace
    nw, nq = sa.n_mesh_w, sa.n_mesh_q
    sr = linspace(sa.w_min, sa.w_max, nw)
    s_yaw = linspace(float(deg2rad(sa.yaw_min)), float(deg2rad(sa.yaw_max)), nq)
    s_pitch = linspace(float(deg2rad(sa.pitch_min)), float(deg2rad(sa.pitch_max)), nq)
    s_roll = linspace(float(deg2rad(sa.roll_min)), float(deg2rad(sa.roll_max)), nq)
    # reshape_states (:717-742): single typed properties
    X1 = X2 = X3 = sr.astype(f32)
    UV = np.asarray(sa.U_vector).astype(f32)
    nu = len(UV)
    C4, S4 = np.cos(s_yaw / 2).astype(f32)[:, None, None], np.sin(s_yaw / 2).astype(f32)[:, None, None]
    C5, S5 = np.cos(s_pitch / 2).astype(f32)[None, :, None], np.sin(s_pitch / 2).astype(f32)[None, :, None]
    C6, S6 = np.cos(s_roll / 2).astype(f32)[None, None, :], np.sin(s_roll / 2).astype(f32)[None, None, :]
    x4 = S4 * C5 * C6 - C4 * S5 * S6
    x5 = C4 * S5 * C6 + S4 * C5 * S6
    x6 = C4 * C5 * S6 - S4 * S5 * C6
    x7 = (f32(1) - (x4 ** 2 + x5 ** 2 + x6 ** 2)) ** f32(0.5)
    h, J1, J2, J3 = f32(sa.h), sa.J1, sa.J2, sa.J3           # double scalar * single array -> single
    A = lambda v: v[:, None, None]
    B = lambda v: v[None, :, None]
    Cv = lambda v: v[None, None, :]
    t1 = h * (f32((J2 - J3) / J1) * A(X2) * B(X3) + Cv(UV) / f32(J1))        # (w2, w3, U1)
    t2 = h * (f32((J3 - J1) / J2) * B(X3) * A(X1) + Cv(UV) / f32(J2))        # (w1, w3, U2)
    t3 = h * (f32((J1 - J2) / J3) * A(X1) * B(X2) + Cv(UV) / f32(J3))        # (w1, w2, U3)
    prob = {"knots": [k.astype(f32).astype(np.float64) for k in (s_yaw, s_pitch, s_roll, sr, sr, sr)],
            "m": [nu, nu, nu], "single": True}
    wterms = [[T(4, X1), T([5, 6, 7], t1)], [T(5, X2), T([4, 6, 8], t2)], [T(6, X3), T([4, 5, 9], t3)]]
    if on_the_fly:
        prob["next_terms"] = [[], [], []] + wterms
        prob["model"] = {"h": float(h), "tables": [x4, x5, x6, x7]}
    else:
        W1, W2, W3 = X1[None, None, None, :, None, None], X2[None, None, None, None, :, None], X3[None, None, None, None, None, :]
        q4, q5, q6, q7 = (a[:, :, :, None, None, None] for a in (x4, x5, x6, x7))
        half = f32(0.5)
        X4n = q4 + h * (half * (W3 * q5 - W2 * q6 + W1 * q7))
        X5n = q5 + h * (half * (-W3 * q4 + W1 * q6 + W2 * q7))
        X6n = q6 + h * (half * (W2 * q4 - W1 * q5 + W3 * q7))
        X7n = q7 + h * (half * (-W1 * q4 - W2 * q5 - W3 * q6))
        nrm = np.sqrt(X4n ** 2 + X5n ** 2 + X6n ** 2 + X7n ** 2)
        X4n, X5n, X6n, X7n = X4n / nrm, X5n / nrm, X6n / nrm, X7n / nrm
        two = f32(2)
        yaw_n = np.arctan2(two * (X6n * X5n + X7n * X4n), X7n ** 2 + X6n ** 2 - X5n ** 2 - X4n ** 2)
        pitch_n = np.arcsin(-two * (X6n * X4n - X7n * X5n))
        roll_n = np.arctan2(two * (X5n * X4n + X7n * X6n), X7n ** 2 - X6n ** 2 - X5n ** 2 + X4n ** 2)
        st6 = [1, 2, 3, 4, 5, 6]
        prob["next_terms"] = [[T(st6, yaw_n.astype(f32))], [T(st6, pitch_n.astype(f32))], [T(st6, roll_n.astype(f32))]] + wterms
    prob["cost_terms"] = [T(4, f32(sa.Q1) * X1 ** 2), T(5, f32(sa.Q2) * X2 ** 2), T(6, f32(sa.Q3) * X3 ** 2),
                          T([1, 2, 3], f32(sa.Q4) * x4 ** 2), T([1, 2, 3], f32(sa.Q5) * x5 ** 2), T([1, 2, 3], f32(sa.Q6) * x6 ** 2),
                          T(7, f32(sa.R1) * UV ** 2), T(8, f32(sa.R2) * UV ** 2), T(9, f32(sa.R3) * UV ** 2)]
    return prob, (nq, nw, nu)


def attitude_run_finish(out, dims):
    """The tail of Solver_attitude_hjbdp_run.m: back to (w1, w2, w3, yaw, pitch, roll), labels split into (i1, i2, i3)."""
    nq, nw, nu = dims
    toref = lambda v: np.transpose(np.asarray(v).reshape([nq, nq, nq, nw, nw, nw], order="F"), (3, 4, 5, 0, 1, 2))
    J = toref(out["J"]).astype(f32)
    lab = toref(out["idx"]) - 1
    i1, i2, i3 = np.mod(lab, nu), np.mod(np.floor(lab / nu), nu), np.floor(lab / (nu * nu))
    return J, (i1.astype(int), i2.astype(int), i3.astype(int))


def pos_att_channel_prob(pa, s_x, s_v, s_t, s_w, f0, f1, f6, f7, Qx, Qv, Qt, Qw, R, J, cost_mode="f64"):
    """matlab/Solver_pos_att_hjbdp_channel.m."""
    from hjbdp.solver_pos_att import vectors_allcomb
    fa, fb, fc, fd = vectors_allcomb(f0, f1, f6, f7)
    h, d = pa.h, pa.T_dist
    dv = h * ((fa + fb + fc + fd) / pa.Mass)
    dw = h * ((fa * d + fb * (-d) + fc * d + fd * (-d)) / J)
    prob = {"knots": [s_x, s_v, s_t, s_w], "m": [len(fa)], "single": True,
            "next_terms": [[T(1, s_x), T(2, h * s_v)], [T(2, s_v), T(5, dv)], [T(3, s_t), T(4, h * s_w)], [T(4, s_w), T(5, dw)]]}
    cu = R * fa ** 2 + R * fb ** 2 + R * fc ** 2 + R * fd ** 2
    if cost_mode == "exact":          # J_current_reshaped (:784-802): single(double sum)
        X, V = s_x[:, None, None, None, None], s_v[None, :, None, None, None]
        Tt, W = s_t[None, None, :, None, None], s_w[None, None, None, :, None]
        full = (Qx * X ** 2 + Qv * V ** 2 + Qw * W ** 2 + Qt * Tt ** 2 + cu[None, None, None, None, :]).astype(f32)
        prob["cost_terms"] = [T([1, 2, 3, 4, 5], full)]
    else:       # 'terms' (summed in single inside the library) and 'f64' (hjbdp_solve 'double_cost': summed in double, one rounding)
        prob["cost_terms"] = [T(1, Qx * s_x ** 2), T(2, Qv * s_v ** 2), T(4, Qw * s_w ** 2), T(3, Qt * s_t ** 2), T(5, cu)]
    return prob, (fa, fb, fc, fd)


POS_ATT_SOLVE_KW = dict(monitor_period=50, monitor_tol=1e-2, monitor_single=True, double_tables=True, labels="auto")
