"""GPU tests (-m gpu) of the parts of the boundary a MATLAB host would use: the flat builder API (primitives and plain
arrays only - what loadlibrary/calllib can marshal) and the probe block (the reference's debug taps,
test/Dynamic_Solver.m:212-219)."""
import ctypes as C

import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import hjbdp
    from hjbdp import _abi
    from oracle import c_oracle
    if hjbdp.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run the HIP path (no fallback)")
    return hjbdp, _abi, c_oracle


@pytest.mark.order(1)
def test_kirk_fixture_through_the_flat_api_only(env, golden):
    """C1a (test/obj_1.txt: 35x35 states x 100 controls, N = 130, float64) solved with nothing but the flat entry
    points, exactly as the MATLAB shim matlab/hjbdp_solve.m drives them: J* within 1e-12 of test/obj_1.mat on all 129
    stages (north_star: 1e-6), u* equal, and equal bit for bit to the struct API."""
    hjbdp, _abi, c_oracle = env
    lib = hjbdp.load_library()
    A = np.array([[0.9974, 0.0539], [-0.1078, 1.1591]])
    B = np.array([0.0013, 0.0539])
    Q11, Q22, R = 0.25, 0.05, 0.05
    s_r = np.ascontiguousarray(golden["knots1"])                 # the fixture's own grid vector (X1_mesh(:,1))
    U = np.ascontiguousarray(golden["U_mesh"])
    dx, du, N = 35, 100, 130
    n = (C.c_int32 * 2)(dx, dx)
    m = (C.c_int32 * 1)(du)
    b = C.c_void_p()

    def ok(st):
        assert st == _abi.HJB_OK, (lib.hjb_problem_last_error(b), lib.hjb_last_error(None))

    ok(lib.hjb_problem_new(2, 1, n, m, _abi.HJB_F64, 1, C.byref(b)))
    for a in range(2):
        ok(lib.hjb_problem_set_knots(b, a, s_r.ctypes.data_as(C.POINTER(C.c_double)), dx))
    # a_D_M (Dynamic_Solver.m:184-188): X_next_a = A(a,1)*X1 + A(a,2)*X2 + B(a)*U, left to right
    for a in range(2):
        for mask, vec in ((0b001, A[a, 0] * s_r), (0b010, A[a, 1] * s_r), (0b100, B[a] * U)):
            v = np.ascontiguousarray(vec)
            ok(lib.hjb_problem_add_next_term(b, a, mask, v.ctypes.data, v.size))
    # g_D (:196-200)
    for mask, vec in ((0b001, Q11 * s_r ** 2), (0b010, Q22 * s_r ** 2), (0b100, R * U ** 2)):
        v = np.ascontiguousarray(vec)
        ok(lib.hjb_problem_add_cost_term(b, mask, v.ctypes.data, v.size))
    h = C.c_void_p()
    ok(lib.hjb_create_from(b, 0, C.byref(h)))
    ok(lib.hjb_problem_free(b))
    nS, n_st = dx * dx, N - 1
    Js = np.zeros((nS, n_st), order="F")
    Is = np.zeros((nS, n_st), dtype=np.int32, order="F")
    Jf = np.zeros(nS)
    If = np.zeros(nS, dtype=np.int32)
    done, early, ms = C.c_int32(), C.c_int32(), C.c_double()
    st = lib.hjb_solve_flat(h, n_st, 0, 0.0, None, Jf.ctypes.data, If.ctypes.data, Js.ctypes.data, Is.ctypes.data,
                            C.byref(done), C.byref(early), C.byref(ms))
    assert st == _abi.HJB_OK, lib.hjb_last_error(h)
    info = (C.c_int64 * 8)()
    assert lib.hjb_get_info_flat(h, info) == _abi.HJB_OK and info[0] == nS and info[1] == du
    assert lib.hjb_destroy(h) == _abi.HJB_OK
    assert done.value == n_st and early.value == 0 and ms.value > 0
    ref = golden["J_star"][:, :, :n_st]
    mine = Js.reshape(dx, dx, n_st, order="F")
    assert np.max(np.abs(mine - ref) / np.abs(ref)) <= 1e-12
    assert np.array_equal(Is.reshape(dx, dx, n_st, order="F") - 1, golden["u_star_idx"][:, :, :n_st])
    assert np.array_equal(Jf, Js[:, 0]) and np.array_equal(If, Is[:, 0])
    # the struct API on the same numbers gives the same bits
    ds = hjbdp.Dynamic_Solver(precision="double")
    ds.N, ds.dx, ds.du = N, dx, du
    with hjbdp.Backup(ds.build_spec()) as bk:
        out = bk.solve(n_st, keep_J=True)
    if np.array_equal(ds.s_r, s_r):
        assert np.array_equal(out["J_stages"], Js)


def test_flat_permute_axes_equals_the_python_relabelling(env):
    """hjb_problem_permute_axes (what a MATLAB host calls after hjb_problem_suggest_order): a problem with terms over
    one, two and three state dims relabelled in C solves to the same bits as the Python relabelling of the same problem
    - and to the original problem's result with J permuted back."""
    hjbdp, _abi, c_oracle = env
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_abi import _builder_from_spec
    from problems import random_problem, random_terminal
    lib = hjbdp.load_library()
    spec = random_problem(21, (6, 5, 7, 4), (3, 2), dtype=np.float32, spread=0.3)
    order = (2, 0, 3, 1)
    pspec, to_old = hjbdp.permute_state_axes(spec, order)
    term = random_terminal(spec, 3)
    pterm = np.transpose(term.reshape(spec.n, order="F"), order).reshape(-1, order="F")
    ref = c_oracle.sweep(_abi, spec, 3, terminal=term)
    b = _builder_from_spec(lib, spec)
    assert lib.hjb_problem_permute_axes(b, (C.c_int32 * 4)(*order)) == _abi.HJB_OK
    h = C.c_void_p()
    assert lib.hjb_create_from(b, 0, C.byref(h)) == _abi.HJB_OK, lib.hjb_problem_last_error(b)
    assert lib.hjb_problem_free(b) == _abi.HJB_OK
    nS = spec.nS
    Jf, If = np.zeros(nS, dtype=np.float32), np.zeros(nS, dtype=np.int32)
    done, early, ms = C.c_int32(), C.c_int32(), C.c_double()
    pt = np.ascontiguousarray(pterm, dtype=np.float32)
    st = lib.hjb_solve_flat(h, 3, 0, 0.0, pt.ctypes.data, Jf.ctypes.data, If.ctypes.data, None, None, C.byref(done), C.byref(early), C.byref(ms))
    assert st == _abi.HJB_OK, lib.hjb_last_error(h)
    lib.hjb_destroy(h)
    with hjbdp.Backup(pspec) as bk:
        out = bk.solve(3, terminal=pterm)
    assert np.array_equal(Jf, out["J"]) and np.array_equal(If, out["idx"])          # C relabelling == Python relabelling
    # the order of the 1-D lerps follows the labelling: against the original problem's sweep equal to rounding only
    assert np.allclose(to_old(Jf), ref["J"], rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("precision", ["single", "double"])
def test_probe_block_equals_the_reference_taps(env, precision):
    """Dynamic_Solver.m:212-219: per stage the sub-block (50:55, 52:57, 105) of J_current_state, X_next_M1, X_next_M2
    (and J_F_next, commented out there).  GPU probe vs the reference's own formulas a_D_M (:184-188) / g_D (:196-200)
    evaluated in numpy with MATLAB's left-to-right order, and vs the interpolation of the oracle."""
    hjbdp, _abi, c_oracle = env
    from oracle import hjb_oracle
    ds = hjbdp.Dynamic_Solver(precision=precision)
    ds.N, ds.dx, ds.du = 6, 60, 110
    ds.run()
    spec = ds.build_spec()
    dt = spec.dtype
    i, j, u = slice(49, 55), slice(51, 57), 104

    def block(terms):
        t0, t1, t2 = (t.data for t in terms)
        return ((t0[i, None] + t1[None, j]).astype(dt) + t2[u]).astype(dt)
    n_st = ds.N - 1
    assert ds.J_current_state_check.shape == (6, 6, n_st) and ds.J_current_state_check.dtype == dt
    for k in range(n_st):
        assert np.array_equal(ds.J_current_state_check[:, :, k], block(spec.cost_terms))
        assert np.array_equal(ds.X_next_M1_check[:, :, k], block(spec.next_terms[0]))
        assert np.array_equal(ds.X_next_M2_check[:, :, k], block(spec.next_terms[1]))
    # against the formulas themselves, in float64
    s_r = ds.s_r.astype(np.float64)
    U = ds._U_mesh[u]
    X1, X2 = np.meshgrid(s_r[i], s_r[j], indexing="ij")
    tol = 1e-6 if precision == "single" else 1e-13
    assert np.allclose(ds.X_next_M1_check[:, :, 2], ds.A[0, 0] * X1 + ds.A[0, 1] * X2 + ds.B[0, 0] * U, rtol=tol)
    assert np.allclose(ds.X_next_M2_check[:, :, 0], ds.A[1, 0] * X1 + ds.A[1, 1] * X2 + ds.B[1, 0] * U, rtol=tol)
    assert np.allclose(ds.J_current_state_check[:, :, 3], ds.Q[0, 0] * X1 ** 2 + ds.Q[1, 1] * X2 ** 2 + ds.R * U ** 2, rtol=tol)
    # J_F_next_check(:,:,k): J_{k+1} interpolated at the tapped next states; loop counter k = 1 taps the terminal cost
    # (zeros), k = 2 taps J of reference stage N-1, ...  (J_star(:,:,N-k+1))
    assert not ds.J_F_next_check[:, :, 0].any()
    knots = [ds.s_r.astype(np.float64)] * 2
    for k in range(1, n_st):
        Jn = ds.J_star[:, :, ds.N - 1 - k].astype(np.float64)            # 0-based plane of reference stage N-k
        q = np.stack([ds.X_next_M1_check[:, :, k].astype(np.float64).ravel(), ds.X_next_M2_check[:, :, k].astype(np.float64).ravel()])
        ref = hjb_oracle.interp_linear(knots, Jn, q).reshape(6, 6)
        assert np.allclose(ds.J_F_next_check[:, :, k], ref, rtol=2e-5 if precision == "single" else 1e-12, atol=1e-6 if precision == "single" else 0)
    # out of range = the reference's index error, as a status
    with hjbdp.Backup(spec) as bk:
        with pytest.raises(hjbdp.HjbError):
            bk.solve(2, probe={"lo": (49, 51), "hi": (61, 57), "control": (104,)})
        with pytest.raises(hjbdp.HjbError):
            bk.solve(2, probe={"lo": (49, 51), "hi": (55, 57), "control": (110,)})
        # one stage from a host J_next (hjb_probe_stage) = the sweep's plane
        one = bk.probe_stage({"lo": (49, 51), "hi": (55, 57), "control": (104,)}, J_next=ds.J_star[:, :, ds.N - 2])
        assert np.array_equal(one["j_interp"], ds.J_F_next_check[:, :, 1])
        assert np.array_equal(one["g"], ds.J_current_state_check[:, :, 0])
    ds35 = hjbdp.Dynamic_Solver(precision=precision)
    ds35.N, ds35.dx, ds35.du = 4, 35, 100
    ds35.run()
    assert ds35.J_current_state_check is None


def test_progress_after_every_stage(env):
    """Dynamic_Solver.m:101 prints one line per stage: hjb_solve_opts.progress_every_stage."""
    hjbdp, _abi, c_oracle = env
    ds = hjbdp.Dynamic_Solver(precision="double")
    ds.N, ds.dx, ds.du = 12, 20, 30
    seen = []
    with hjbdp.Backup(ds.build_spec()) as bk:
        out = bk.solve(11, progress=lambda k_s, e, e2, sec: seen.append((k_s, sec)), progress_every_stage=True)
        ref = bk.solve(11)
    assert [k for k, _ in seen] == list(range(11, 0, -1))
    assert all(b[1] >= a[1] for a, b in zip(seen, seen[1:]))
    assert np.array_equal(out["J"], ref["J"])


@pytest.mark.parametrize("case", ["colsweep", "nested3", "monitor", "f16"])
def test_solve_multi_matches_single_device(env, case):
    """hjb_create_multi / hjb_solve_multi (the single-process multi-GPU sweep a MATLAB host would call): the grid in
    several slabs with per-stage device-to-device halo copies overlapped with the interior planes.  The 1-GPU box
    gives every slab the same device (the copies are then plain device-to-device); bit-equal to the oracle's whole
    grid sweep, including the early-stop monitor's decision."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, nested_problem, random_terminal
    kw = {}
    if case == "colsweep":        # window axis last: halo of one plane; interior + strips
        spec, devs, stages = colsweep_problem(5, (36, 7, 9, 14), gax=2), [0, 0, 0], 5
    elif case == "nested3":
        spec, devs, stages = nested_problem(31, (7, 6, 13), (3, 4), dtype=np.float32, spread=0.12), [0, 0], 4
    elif case == "monitor":
        spec, devs, stages = colsweep_problem(6, (20, 6, 8, 12), gax=2, cost="multi"), [0, 0], 11
        kw = {"monitor_period": 3, "monitor_tol": 1e12}      # stops at the first monitor point, k_s = 9, after 3 stages
    else:
        spec, devs, stages = colsweep_problem(7, (33, 6, 8, 15), gax=2, j_storage=np.float16), [0, 0, 0, 0], 3
    term = random_terminal(spec, 4)
    ref = c_oracle.sweep(_abi, spec, stages, terminal=term, keep_J=True, keep_idx=True, **kw)
    with hjbdp.MultiBackup(spec, devs) as mb:
        infos = [mb.slab_info(i) for i in range(len(devs))]
        out = mb.solve(stages, terminal=term, keep_J=True, keep_idx=True, **kw)
        again = mb.solve(stages, terminal=term, **kw)            # a second sweep on the same object
    assert infos[0]["begin"] == 0 and infos[-1]["end"] == spec.n[-1]
    assert all(a["end"] == b["begin"] for a, b in zip(infos, infos[1:]))
    if case != "nested3":
        assert any(i["split"] for i in infos), infos          # the overlapped form was exercised
    assert out["stages_done"] == ref["stages_done"] and out["stopped_early"] == ref["stopped_early"]
    assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
    assert np.array_equal(out["J_stages"], ref["J_stages"]) and np.array_equal(out["idx_stages"], ref["idx_stages"])   # every stage's planes
    assert np.array_equal(again["J"], ref["J"]) and np.array_equal(again["idx"], ref["idx"])
    if kw:
        assert out["stopped_early"] and out["stages_done"] == 3
        assert abs(out["last_e"] - ref["last_e"]) <= 1e-9 * abs(ref["last_e"])


def test_solve_multi_refuses_what_it_cannot_do(env):
    hjbdp, _abi, c_oracle = env
    from problems import random_problem
    spec = random_problem(5, (6, 5, 16), (3,), dtype=np.float32, spread=0.6)      # last axis moves several planes
    with pytest.raises(hjbdp.HjbError):
        hjbdp.MultiBackup(spec, [0] * 8)                      # halo wider than a 2-plane slab
    with pytest.raises(hjbdp.HjbError):
        hjbdp.MultiBackup(spec, [0] * 17)                     # more slabs than planes


@pytest.mark.parametrize("overlap", [True, False])
def test_rank_api_three_ranks_in_one_process(env, overlap):
    """hjb_rank_create / hjb_rank_stage (one process per GPU; here three ranks driven from one process, all on this box's
    GPU): library-owned buffers, the halo planes moved by the caller with device-to-device copies between the ranks'
    buffers before each stage - the role MPI / RCCL play in a real run - and every stage one hjb_rank_stage call per rank
    (interior + strips with overlap).  All planes of every stage equal the oracle's whole-grid sweep."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, random_terminal
    spec0 = colsweep_problem(15, (36, 7, 9, 14), gax=2)
    spec = hjbdp.ProblemSpec(spec0.knots, spec0.m, spec0.next_terms, spec0.cost_terms, dtype=np.float32, index_base=1, idx_dtype="auto")
    term = random_terminal(spec, 8)
    stages, world = 4, 3
    ref = c_oracle.sweep(_abi, spec, stages, terminal=term, keep_J=True, keep_idx=True)
    inner = spec.nS // spec.n[-1]
    ranks = [hjbdp.RankSlab(spec, 0, r, world, overlap=overlap) for r in range(world)]
    assert ranks[0].begin == 0 and ranks[-1].end == spec.n[-1] and all(a.end == b.begin for a, b in zip(ranks, ranks[1:]))
    assert any(r.split for r in ranks) == overlap
    lib = ranks[0].lib
    T = term.reshape(inner, -1, order="F")
    bufs = []
    for r in ranks:
        planes = r.end - r.begin + r.halo_lo + r.halo_hi
        J = [hjbdp.DeviceBuffer(inner * planes * 4), hjbdp.DeviceBuffer(inner * planes * 4)]
        init = np.zeros((inner, planes), dtype=np.float32, order="F")
        init[:, r.halo_lo:r.halo_lo + r.end - r.begin] = T[:, r.begin:r.end]
        J[0].upload(init.reshape(-1, order="F"))
        bufs.append((J, hjbdp.DeviceBuffer(inner * (r.end - r.begin) * r.idx_bytes)))
    pb = inner * 4

    def d2d(dst, dst_plane, src, src_plane, n):
        st = lib.hjb_device_copy(0, dst.ptr + pb * dst_plane, src.ptr + pb * src_plane, pb * n, _abi.HJB_COPY_D2D)
        assert st == 0
    cur = 0
    for k_s in range(stages, 0, -1):
        for i, r in enumerate(ranks):                     # the exchange: my halo planes from my neighbours' owned planes
            if r.halo_lo:
                lo = ranks[i - 1]
                d2d(bufs[i][0][cur], 0, bufs[i - 1][0][cur], lo.halo_lo + (lo.end - lo.begin) - r.halo_lo, r.halo_lo)
            if r.halo_hi:
                hi = ranks[i + 1]
                d2d(bufs[i][0][cur], r.halo_lo + r.end - r.begin, bufs[i + 1][0][cur], hi.halo_lo, r.halo_hi)
        for i, r in enumerate(ranks):
            r.stage(bufs[i][0][cur], bufs[i][0][cur ^ 1], bufs[i][1])
        for i, r in enumerate(ranks):
            r.check_device_status()
            planes = r.end - r.begin + r.halo_lo + r.halo_hi
            Jr = bufs[i][0][cur ^ 1].download(np.float32).reshape(inner, planes, order="F")[:, r.halo_lo:r.halo_lo + r.end - r.begin]
            Ir = bufs[i][1].download(spec.idx_np_dtype).reshape(inner, r.end - r.begin, order="F")
            want_J = ref["J_stages"][:, k_s - 1].reshape(inner, -1, order="F")[:, r.begin:r.end]
            want_I = ref["idx_stages"][:, k_s - 1].reshape(inner, -1, order="F")[:, r.begin:r.end]
            assert np.array_equal(Jr, want_J) and np.array_equal(Ir, want_I), (k_s, i)
        cur ^= 1
    for r in ranks:
        r.close()
    with pytest.raises(hjbdp.HjbError):
        hjbdp.RankSlab(spec, 0, 3, 3)                     # rank out of range


def test_flat_api_reference_typing_of_pos_att(env):
    """The flat call sequence a MATLAB host makes for Solver_pos_att in the reference's typing (hjbdp_solve.m with
    'double_tables'): hjb_problem_set_types(AUTO, HJB_TAB_F64), next-state operands handed over as doubles, labels back as
    one byte per state, the monitor in single precision through the handle option - equal to the oracle bit for bit."""
    hjbdp, _abi, c_oracle = env
    lib = hjbdp.load_library()
    pa = hjbdp.Solver_pos_att()
    pa.n_mesh_x, pa.n_mesh_v, pa.n_mesh_t, pa.n_mesh_w = 14, 9, 8, 7
    pa.cost_mode = "terms"                  # single cost operands (this test is about the TABLE typing; the double cost: test_gpu_types.py)
    sx, sv, st, sw = pa.grids()
    spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    assert spec.table_dtype == np.float64 and spec.idx_np_dtype == np.uint8 and spec.cost_dtype is None
    b = C.c_void_p()
    n = (C.c_int32 * 4)(*spec.n)
    m = (C.c_int32 * 1)(*spec.m)
    assert lib.hjb_problem_new(4, 1, n, m, _abi.HJB_F32, 1, C.byref(b)) == 0
    assert lib.hjb_problem_set_types(b, _abi.HJB_IDX_AUTO, _abi.HJB_TAB_F64) == 0
    for a in range(4):
        k = np.ascontiguousarray(spec.knots[a])
        assert lib.hjb_problem_set_knots(b, a, k.ctypes.data_as(C.POINTER(C.c_double)), k.size) == 0
        for t in spec.next_terms[a]:
            v = np.ascontiguousarray(np.asarray(t.data, dtype=np.float64).reshape(-1, order="F"))       # doubles
            assert lib.hjb_problem_add_next_term(b, a, sum(1 << d for d in t.dims), v.ctypes.data, v.size) == 0
    for t in spec.cost_terms:
        v = np.ascontiguousarray(np.asarray(t.data, dtype=np.float32).reshape(-1, order="F"))           # singles
        assert lib.hjb_problem_add_cost_term(b, sum(1 << d for d in t.dims), v.ctypes.data, v.size) == 0
    h = C.c_void_p()
    assert lib.hjb_create_from(b, 0, C.byref(h)) == 0, lib.hjb_problem_last_error(b)
    lib.hjb_problem_free(b)
    assert lib.hjb_set_option(h, b"monitor_single", 1) == 0
    stages = 30
    J = np.empty(spec.nS, dtype=np.float32)
    idx = np.empty(spec.nS, dtype=np.uint8)
    done, early, ms = C.c_int32(), C.c_int32(), C.c_double()
    st_ = lib.hjb_solve_flat(h, stages, 10, 1e-2, None, J.ctypes.data, idx.ctypes.data, None, None, C.byref(done), C.byref(early), C.byref(ms))
    assert st_ == 0, lib.hjb_last_error(h)
    lib.hjb_destroy(h)
    ref = c_oracle.sweep(_abi, spec, stages, monitor_period=10, monitor_tol=1e-2, monitor_single=True)
    assert done.value == ref["stages_done"] and bool(early.value) == ref["stopped_early"]
    assert np.array_equal(J, ref["J"]) and np.array_equal(idx, ref["idx"])


# ---- the MATLAB solver shims' call sequences (matlab/*.m), replayed through ctypes: tests/matlab_twin.py ----------------

def _twin():
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import matlab_twin
    return matlab_twin


@pytest.mark.order(8)
def test_matlab_shim_sequences_solver_position(env):
    """matlab/Solver_position_hjbdp_simplified_run.m: the three channels through hjbdp_solve.m's call sequence equal the
    Python mirror (oracle-checked in tests/test_gpu_solvers.py) bit for bit, on the reference's own grid; the 'nearest'
    policy tables U_vector(U_idx) follow."""
    hjbdp, _abi, c_oracle = env
    mt = _twin()
    lib = hjbdp.load_library()
    sp = hjbdp.Solver_position()
    sp.simplified_run(n_stages=40, keep_policy=True)
    assert sp.U_Opt_stages[0].shape == (201, 201, 40) and np.array_equal(sp.U_idx_stages[1][:, :, 0], sp.U_idx[1])
    orc = c_oracle.sweep(_abi, hjbdp.Solver_position().build_spec(2)[0], 40, keep_idx=True)     # a fresh object: the run updates n_mesh_x / n_mesh_v (:100,:104)
    assert np.array_equal(sp.U_idx_stages[2].reshape(-1, 40, order="F"), orc["idx_stages"])
    ref = hjbdp.Solver_position()
    for ch in range(3):
        prob = mt.position_channel_prob(ref, ch)
        assert len(prob["knots"][0]) == 201 and len(prob["knots"][1]) == 201           # sym_linspace: 200 -> 201 points
        out = mt.hjbdp_solve(lib, prob, 40)
        assert out["J"].shape == (201, 201) and out["idx"].dtype == np.float64
        assert np.array_equal(out["J"], sp.F_values[ch]) and np.array_equal(out["idx"], sp.U_idx[ch])
        pol = ref.U_vector[out["idx"].astype(int) - 1]                                  # griddedInterpolant(..., U_vector(U_idx), 'nearest')
        assert np.array_equal(pol, getattr(sp, "U%d_Opt" % (ch + 1)).Values)
    # ... and on two "devices" of this process (hjb_create_multi_from / hjb_solve_multi_flat)
    out2 = mt.hjbdp_solve(lib, mt.position_channel_prob(ref, 0), 40, devices=[0, 0])
    assert np.array_equal(out2["J"], sp.F_values[0]) and np.array_equal(out2["idx"], sp.U_idx[0])
    # ... and with the caller keeping its own `for k` loop on device buffers (hjbdp_solve.m 'on_stage': hjb_device_malloc /
    # hjb_backup_stage_device / hjb_check_device_status / hjb_device_copy): the same bits, every stage seen, early stop honoured
    seen = []
    out3 = mt.hjbdp_solve(lib, mt.position_channel_prob(ref, 0), 40, on_stage=lambda k: seen.append(k) or False)
    assert seen == list(range(40, 0, -1)) and out3["stages_done"] == 40 and not out3["stopped_early"]
    assert np.array_equal(out3["J"], sp.F_values[0]) and np.array_equal(out3["idx"], sp.U_idx[0])
    sp7 = hjbdp.Solver_position()
    sp7.simplified_run(n_stages=7)
    out4 = mt.hjbdp_solve(lib, mt.position_channel_prob(ref, 0), 40, on_stage=lambda k: k == 34)       # stages 40 .. 34: seven
    assert out4["stages_done"] == 7 and out4["stopped_early"]
    assert np.array_equal(out4["J"], sp7.F_values[0]) and np.array_equal(out4["idx"], sp7.U_idx[0])


@pytest.mark.order(8)
def test_matlab_shim_sequences_attitude_simplified(env):
    """matlab/Solver_attitude_hjbdp_simplified_run.m against the Python mirror."""
    hjbdp, _abi, c_oracle = env
    mt = _twin()
    lib = hjbdp.load_library()
    sa = hjbdp.Solver_attitude(n_mesh_t=60, n_mesh_w_simplified=140)
    sa.simplified_run(n_stages=30, keep_policy=True)
    plain = hjbdp.Solver_attitude(n_mesh_t=60, n_mesh_w_simplified=140).simplified_run(n_stages=30)
    assert plain.U_Opt_stages is None
    for ch in range(3):
        out = mt.hjbdp_solve(lib, mt.attitude_simplified_prob(sa, ch), 30, keep_stages=True, fast_axes=False)     # the shim's 'keep_policy' (the mirror runs the reference's axis order)
        assert np.array_equal(out["J"], sa.F_values[ch]) and np.array_equal(out["idx"], sa.U_idx[ch])
        assert np.array_equal(plain.F_values[ch], sa.F_values[ch]) and np.array_equal(plain.U_idx[ch], sa.U_idx[ch])
        # every stage's policy (attitude-control/test/test_simplified.m:102-104): U_vector(U_idx) per stage, page k_s - 1
        pol = sa.U_vector[out["idx_stages"].astype(int) - 1].reshape(140, 60, 30, order="F")
        assert sa.U_Opt_stages[ch].shape == (140, 60, 30) and np.array_equal(pol, sa.U_Opt_stages[ch])
        assert np.array_equal(sa.U_idx_stages[ch][:, :, 0], sa.U_idx[ch])                      # the last stage computed is k_s = 1
        spec = sa.build_spec_simplified(ch)[0]
        ref = c_oracle.sweep(_abi, spec, 30, keep_idx=True)
        assert np.array_equal(sa.U_idx_stages[ch].reshape(-1, 30, order="F"), ref["idx_stages"])


@pytest.mark.order(8)
@pytest.mark.parametrize("on_the_fly", [True, False])
def test_matlab_shim_sequences_attitude_run(env, on_the_fly):
    """matlab/Solver_attitude_hjbdp_run.m: the 6-D problem built in the library's axis order with hjb_problem_set_model
    (on the fly) or with the three tabulated next-angle operands, uint8 labels ('labels', 'auto'), results permuted back:
    obj.F.Values and the three U_i_Opt index arrays equal the Python mirror's run()."""
    hjbdp, _abi, c_oracle = env
    mt = _twin()
    lib = hjbdp.load_library()
    sa = hjbdp.Solver_attitude(n_mesh_w=6, n_mesh_q=5)
    sa.run(n_stages=5, on_the_fly=on_the_fly)
    assert sa.kernel_variant == 4
    prob, dims = mt.attitude_run_prob(sa, on_the_fly=on_the_fly)
    out = mt.hjbdp_solve(lib, prob, 5, labels="auto")
    J, (i1, i2, i3) = mt.attitude_run_finish(out, dims)
    assert J.shape == (6, 6, 6, 5, 5, 5)
    assert np.array_equal(J, sa.F_values)
    for mine, theirs in zip((i1, i2, i3), sa.U_idx):
        assert np.array_equal(mine + 1, theirs)
    assert np.array_equal(sa.U_vector[i3].astype(np.float32), sa.U3_Opt)


@pytest.mark.order(8)
@pytest.mark.parametrize("cost_mode", ["exact", "terms", "f64"])
def test_matlab_shim_sequences_pos_att_channel(env, cost_mode):
    """matlab/Solver_pos_att_hjbdp_channel.m on the reference's own grid (30 x 30 x 20 x 15 x 9): double query tables,
    single-precision monitor every 50 stages, uint8 labels - F_gI.Values, U_Optimal_id, the stage the monitor stopped at
    and the *_allcomb vectors equal the Python mirror's channel (oracle-checked incl. the monitor); with 'fast_axes' the
    library's relabelling gives the same J to rounding."""
    hjbdp, _abi, c_oracle = env
    mt = _twin()
    lib = hjbdp.load_library()
    pa = hjbdp.Solver_pos_att()
    pa.cost_mode = cost_mode
    sx, sv, st, sw = pa.grids()
    args = (sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    prob, combos = mt.pos_att_channel_prob(pa, *args, cost_mode=cost_mode)
    outs = {}
    # the shim's default (hjbdp_solve's 'fast_axes' true = the mirror's axis_order "auto") and the reference's own axis order
    for fast in ((True, False) if cost_mode != "exact" else (False,)):
        pa.axis_order = "auto" if fast else None
        c = pa.calculate_one_channel_U_Opt(*args, "channel_x_controller_1", n_stages=120)
        out = mt.hjbdp_solve(lib, prob, 120, double_cost=(cost_mode == "f64"), fast_axes=fast, **mt.POS_ATT_SOLVE_KW)
        assert out["J"].shape == (30, 30, 20, 15)
        assert out["axis_order"] == ([1, 3, 4, 2] if fast else [1, 2, 3, 4])
        assert np.array_equal(out["J"], c["F_gI_Values"]) and np.array_equal(out["idx"], c["U_Optimal_id"]), fast
        assert out["stages_done"] == c["stages_done"] and out["stopped_early"] == c["stopped_early"]
        for k, v in zip(("f0_allcomb", "f1_allcomb", "f6_allcomb", "f7_allcomb"), combos):
            assert np.array_equal(v, c[k])
        outs[fast] = out
    if True in outs:
        assert np.allclose(outs[True]["J"], outs[False]["J"], rtol=5e-5, atol=1e-6)


def test_rank_create_from_builder_equals_struct_form(env):
    """hjb_rank_create_from (the flat form a MATLAB worker per GPU binds): same partition, same stage kernel, same bits as
    hjb_rank_create on the struct."""
    hjbdp, _abi, c_oracle = env
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_abi import _builder_from_spec
    from problems import colsweep_problem, random_terminal
    lib = hjbdp.load_library()
    spec = colsweep_problem(90, (66, 8, 9, 12), nU=9)
    term = random_terminal(spec, 2)
    Jr, ir = c_oracle.backup_stage(_abi, spec, term)
    nl, inner = spec.n[-1], spec.nS // spec.n[-1]
    for rank in range(3):
        b = _builder_from_spec(lib, spec)
        r = C.c_void_p()
        assert lib.hjb_rank_create_from(b, 0, rank, 3, 1, C.byref(r)) == _abi.HJB_OK, lib.hjb_problem_last_error(b)
        assert lib.hjb_problem_free(b) == _abi.HJB_OK
        info = (C.c_int32 * 10)()
        assert lib.hjb_rank_info(r, info) == _abi.HJB_OK
        begin, end, hlo, hhi = info[0], info[1], info[2], info[3]
        rb = hjbdp.core.RankSlab(spec, 0, rank, 3)
        assert (rb.begin, rb.end, rb.halo_lo, rb.halo_hi, rb.split, rb.kernel_variant) == (begin, end, hlo, hhi, info[4], info[5])
        rb.close()
        planes = end - begin + hlo + hhi
        view = np.ascontiguousarray(term.reshape(inner, nl, order="F")[:, begin - hlo:end + hhi].reshape(-1, order="F"))
        with hjbdp.DeviceBuffer(view.nbytes) as dIn, hjbdp.DeviceBuffer(view.nbytes) as dOut, \
                hjbdp.DeviceBuffer(inner * (end - begin) * info[8]) as dI:
            dIn.upload(view)
            dOut.upload(view)
            assert lib.hjb_rank_stage(r, int(dIn), int(dOut), int(dI), None, None) == _abi.HJB_OK, lib.hjb_rank_last_error(r)
            assert lib.hjb_rank_check_status(r, None) == _abi.HJB_OK
            Jo = dOut.download(np.float32).reshape(inner, planes, order="F")[:, hlo:hlo + end - begin]
            io = dI.download({1: np.uint8, 2: np.uint16, 4: np.int32}[info[8]])
        assert np.array_equal(Jo, Jr.reshape(inner, nl, order="F")[:, begin:end])
        assert np.array_equal(io.reshape(inner, end - begin, order="F"), ir.reshape(inner, nl, order="F")[:, begin:end])
        assert lib.hjb_rank_destroy(r) == _abi.HJB_OK


def _rccl_unique_id(lib, _abi):
    """The 128-byte id of a new communicator, or skip: a host without librccl gets HJB_E_UNSUPPORTED (include/hjbdp.h)."""
    uid = (C.c_char * 128)()
    st = lib.hjb_rank_comm_unique_id(uid)
    if st == _abi.HJB_E_UNSUPPORTED:
        pytest.skip("RCCL is not available here: %s" % (lib.hjb_rank_last_error(None) or b"").decode())
    assert st == _abi.HJB_OK, lib.hjb_rank_last_error(None)
    assert lib.hjb_rank_comm_available() == _abi.HJB_OK             # the side-effect-free probe agrees
    return uid


@pytest.mark.parametrize("overlap", [True, False])
def test_rccl_transport_inside_the_library_loopback(env, overlap):
    """hjb_rank_comm_init / hjb_rank_exchange / hjb_rank_step / hjb_rank_monitor_sums: the RCCL calls of a middle rank
    (ncclGroupStart, two ncclSend, two ncclRecv, ncclGroupEnd on the library's transfer stream; a 2-double ncclAllReduce)
    executed for real on this box's ONE GPU through the loopback option - a communicator of one rank that is both of its
    neighbours: the planes it sends down arrive in its own upper halo, those it sends up in its lower halo.  After the
    exchange the halos hold exactly those planes, and the stage that follows (interior beside the transfer, strips behind
    it) equals the oracle's backup of that buffer bit for bit."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, random_terminal
    spec0 = colsweep_problem(15, (36, 7, 9, 14), gax=2)
    spec = hjbdp.ProblemSpec(spec0.knots, spec0.m, spec0.next_terms, spec0.cost_terms, dtype=np.float32, index_base=1, idx_dtype="auto")
    inner = spec.nS // spec.n[-1]
    rk = hjbdp.core.RankSlab(spec, 0, 1, 3, overlap=overlap)
    lib = rk.lib
    assert rk.halo_lo > 0 and rk.halo_hi > 0 and rk.split == (1 if overlap else 0)
    owned, hlo, hhi = rk.end - rk.begin, rk.halo_lo, rk.halo_hi
    planes = owned + hlo + hhi
    rk.set_option("comm_loopback", 1)
    uid = _rccl_unique_id(lib, _abi)
    assert lib.hjb_rank_comm_init(rk._r, uid) == _abi.HJB_OK, lib.hjb_rank_last_error(rk._r)
    assert lib.hjb_rank_transfer_stream(rk._r)
    nr, me = C.c_int32(-5), C.c_int32(-5)
    assert lib.hjb_rank_comm_info(rk._r, C.byref(nr), C.byref(me)) == _abi.HJB_OK      # the loopback communicator: one rank, rank 0
    assert (nr.value, me.value) in ((1, 0), (-1, -1))
    rng = np.random.default_rng(5)
    init = (rng.random((inner, planes)) * 3).astype(np.float32)
    with hjbdp.DeviceBuffer(init.nbytes) as dIn, hjbdp.DeviceBuffer(init.nbytes) as dOut, \
            hjbdp.DeviceBuffer(inner * owned * rk.idx_bytes) as dI:
        dIn.upload(np.asfortranarray(init).reshape(-1, order="F"))
        dOut.upload(np.asfortranarray(init).reshape(-1, order="F"))
        assert lib.hjb_rank_exchange(rk._r, int(dIn), None) == _abi.HJB_OK, lib.hjb_rank_last_error(rk._r)
        rk.check_device_status(lib.hjb_rank_transfer_stream(rk._r))          # synchronises the transfer stream
        got = dIn.download(np.float32).reshape(inner, planes, order="F")
        want = init.copy()
        want[:, :hlo] = init[:, hlo + owned - hlo:hlo + owned]               # sent "up", came back as my lower halo
        want[:, hlo + owned:] = init[:, hlo:hlo + hhi]                       # sent "down", came back as my upper halo
        assert np.array_equal(got, want)
        # exchange + stage in one call; the oracle backs the exchanged buffer up as this rank's slab
        dIn.upload(np.asfortranarray(init).reshape(-1, order="F"))
        assert lib.hjb_rank_step(rk._r, int(dIn), int(dOut), int(dI), None) == _abi.HJB_OK, lib.hjb_rank_last_error(rk._r)
        rk.check_device_status()
        Jo, io = c_oracle.backup_stage(_abi, spec, np.asfortranarray(want).reshape(-1, order="F"), slab=(rk.begin, rk.end, hlo, hhi))
        mine = dOut.download(np.float32).reshape(inner, planes, order="F")[:, hlo:hlo + owned]
        assert np.array_equal(mine, Jo.reshape(inner, planes, order="F")[:, hlo:hlo + owned])
        assert np.array_equal(dI.download(spec.idx_np_dtype), io)
        # the monitor's two sums through ncclAllReduce (one rank: the local sums)
        sums = (C.c_double * 2)()
        assert lib.hjb_rank_monitor_sums(rk._r, int(dOut), int(dI), None, sums) == _abi.HJB_OK, lib.hjb_rank_last_error(rk._r)
        assert abs(sums[0] - float(mine.astype(np.float64).sum())) <= 1e-9 * abs(sums[0])
        assert sums[1] == float(io.astype(np.float64).sum())
        # hjb_rank_step_post: the strips first, the exchange of the OUTPUT's boundary planes behind the strips, under the interior.
        # One exchange of the input, then the step: the stage equals the oracle's backup of the exchanged input, and the output
        # comes back with its own halos filled (loopback: its own boundary planes)
        dIn.upload(np.asfortranarray(init).reshape(-1, order="F"))
        dOut.upload(np.full(init.size, -7.0, dtype=np.float32))
        assert lib.hjb_rank_exchange(rk._r, int(dIn), None) == _abi.HJB_OK, lib.hjb_rank_last_error(rk._r)
        assert lib.hjb_rank_step_post(rk._r, int(dIn), int(dOut), int(dI), None) == _abi.HJB_OK, lib.hjb_rank_last_error(rk._r)
        rk.check_device_status()
        rk.check_device_status(lib.hjb_rank_transfer_stream(rk._r))
        got = dOut.download(np.float32).reshape(inner, planes, order="F")
        own = Jo.reshape(inner, planes, order="F")[:, hlo:hlo + owned]
        assert np.array_equal(got[:, hlo:hlo + owned], own) and np.array_equal(dI.download(spec.idx_np_dtype), io)
        assert np.array_equal(got[:, :hlo], own[:, owned - hlo:]) and np.array_equal(got[:, hlo + owned:], own[:, :hhi])
        # the whole loop both ways (option "post_exchange"): the same sweep, equal to the oracle driven stage by stage with the
        # loopback's halos (each stage's input halos = that buffer's own boundary planes)
        cur = init.copy()
        for _ in range(5):
            cur[:, :hlo] = cur[:, owned:owned + hlo]
            cur[:, hlo + owned:] = cur[:, hlo:hlo + hhi]
            nxt, iref = c_oracle.backup_stage(_abi, spec, np.asfortranarray(cur).reshape(-1, order="F"), slab=(rk.begin, rk.end, hlo, hhi))
            cur = nxt.reshape(inner, planes, order="F").copy()
        for post in (1, 0):
            rk.set_option("post_exchange", post)
            dIn.upload(np.asfortranarray(init).reshape(-1, order="F"))
            dOut.upload(np.asfortranarray(init).reshape(-1, order="F"))
            done, early, in0, ms = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
            st = lib.hjb_rank_sweep(rk._r, 5, 0, 0.0, int(dIn), int(dOut), int(dI), None, C.byref(done), C.byref(early), C.byref(in0), C.byref(ms))
            assert st == _abi.HJB_OK and done.value == 5, lib.hjb_rank_last_error(rk._r)
            rk.check_device_status(lib.hjb_rank_transfer_stream(rk._r))
            J = (dIn if in0.value else dOut).download(np.float32).reshape(inner, planes, order="F")
            assert np.array_equal(J[:, hlo:hlo + owned], cur[:, hlo:hlo + owned]), post
            assert np.array_equal(dI.download(spec.idx_np_dtype), iref), post
    rk.close()


def test_rank_sweep_whole_loop_in_the_library(env):
    """hjb_rank_sweep (what tools/bench_ranks.cpp and a MATLAB worker per GPU call): the `for k_s` loop of one rank with
    the monitor's all-reduced sums; at world = 1 it must be hjb_solve's sweep - same J, same labels, same stop stage."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, random_terminal
    spec0 = colsweep_problem(16, (40, 7, 8, 9), gax=3)
    spec = hjbdp.ProblemSpec(spec0.knots, spec0.m, spec0.next_terms, spec0.cost_terms, dtype=np.float32, index_base=1, idx_dtype="auto")
    term = random_terminal(spec, 4)
    rk = hjbdp.core.RankSlab(spec, 0, 0, 1)
    lib = rk.lib
    uid = _rccl_unique_id(lib, _abi)
    assert lib.hjb_rank_comm_init(rk._r, uid) == _abi.HJB_OK, lib.hjb_rank_last_error(rk._r)
    assert lib.hjb_rank_comm_init(rk._r, uid) == _abi.HJB_E_INVALID          # one communicator per rank
    # the reference's typing of the monitor (Solver_pos_att.m:274-282: single sum, single difference, single comparison) at
    # the margin: a tolerance one float32 ulp above the second monitor point's |e| stops there, |e| itself does not ('<')
    events = []
    with hjbdp.Backup(spec) as bk:
        bk.solve(9, terminal=term, monitor_period=3, monitor_tol=0.0, monitor_single=True, progress=lambda k, e, e2, sec: events.append(e))
    e2nd = np.float32(abs(events[1]))
    cases = [(0.0, False), (1e30, False), (float(np.nextafter(e2nd, np.float32(np.inf))), True), (float(e2nd), True)]
    for tol, single in cases:                                 # never stops / stops at the first monitor point / the margin
        rk.set_option("monitor_single", 1 if single else 0)
        ref = c_oracle.sweep(_abi, spec, 9, terminal=term, monitor_period=3, monitor_tol=tol, monitor_single=single)
        with hjbdp.DeviceBuffer(spec.nS * 4) as d0, hjbdp.DeviceBuffer(spec.nS * 4) as d1, hjbdp.DeviceBuffer(spec.nS * rk.idx_bytes) as dI:
            d0.upload(term)
            done, early, in0, ms = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
            st = lib.hjb_rank_sweep(rk._r, 9, 3, tol, int(d0), int(d1), int(dI), None, C.byref(done), C.byref(early), C.byref(in0), C.byref(ms))
            assert st == _abi.HJB_OK, lib.hjb_rank_last_error(rk._r)
            assert (done.value, bool(early.value)) == (ref["stages_done"], ref["stopped_early"])
            J = (d0 if in0.value else d1).download(np.float32)
            assert np.array_equal(J, ref["J"]) and np.array_equal(dI.download(spec.idx_np_dtype), ref["idx"])
            assert ms.value > 0
    rk.close()


@pytest.mark.watchdog(600)
def test_cpp_rank_driver_matches_the_python_path(env):
    """tools/bench_ranks.cpp (SURVEY 8b iii: a C++ driver, one process per GPU, RCCL inside the library, no Python in the
    loop): built here with g++ against the in-tree libhjbdp.so and run as one rank on a 34^4 pos-att grid.  It states the
    problem in C++ through the flat builder; the sum of J after 1 + 12 stages must equal hjb_solve's on bench.py's spec of
    the same grid (the same operands bit for bit, or the sums would differ), and the JSON line carries bench.py's keys."""
    import json
    import subprocess
    hjbdp, _abi, c_oracle = env
    import bench
    _rccl_unique_id(hjbdp.load_library(), _abi)                # skips on a host without librccl
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run(["bash", str(root / "tools" / "build_bench_ranks.sh")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    run = subprocess.run([str(root / "tools" / "bench_ranks"), "--gpus", "1", "--steps", "12", "--warmup", "1", "--grid-n", "34"],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-2000:])
    line = json.loads(run.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "checksum_sum_J"):
        assert key in line, key
    assert line["metric"] == "bellman_backups_per_s" and line["n_gpus"] == 1 and line["steps"] == 12
    spec, _ = bench.build_spec("c4", n=34)
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["kernel_variant"] == line["config"]["kernel_variant"]        # the library's own choice on both sides
        out = bk.solve(13)
    want = float(out["J"].astype(np.float64).sum())
    assert abs(line["checksum_sum_J"] - want) <= 1e-11 * abs(want), (line["checksum_sum_J"], want)
    assert abs(line["value"] - 34 ** 4 * 9 * 12 / (line["ms_per_step"] * 12e-3)) <= 1e-4 * line["value"]      # both are printed rounded
