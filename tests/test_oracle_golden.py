"""CPU tests: the oracle against the reference's golden vector (test/obj_1.mat,
committed as tests/golden/obj_1.npz) and the two oracle implementations against
each other.  Tolerances: float64 1e-12 relative (observed 7e-14); the float32
restatement is only required to track float64 to 1e-4 (observed 2.6e-5)."""
import numpy as np
import pytest

from problems import random_problem, random_terminal


@pytest.fixture(scope="module")
def orc(built):
    from hjbdp import _abi
    from oracle import c_oracle, hjb_oracle
    return _abi, c_oracle, hjb_oracle


def kirk_spec(golden, dtype=np.float64):
    from hjbdp import ProblemSpec, Term
    A, B, Q, R = golden["A"], golden["B"], golden["Q"], float(golden["R"])
    k1, k2, U = golden["knots1"], golden["knots2"], golden["U_mesh"]
    nxt = [[Term((0,), A[0, 0] * k1), Term((1,), A[0, 1] * k2), Term((2,), B[0, 0] * U)],
           [Term((0,), A[1, 0] * k1), Term((1,), A[1, 1] * k2), Term((2,), B[1, 0] * U)]]
    cost = [Term((0,), Q[0, 0] * k1 ** 2), Term((1,), Q[1, 1] * k2 ** 2), Term((2,), R * U ** 2)]
    return ProblemSpec([k1, k2], [len(U)], nxt, cost, dtype=dtype)


def test_numpy_oracle_reproduces_matlab_fixture(orc, golden):
    _abi, c_oracle, hjb_oracle = orc
    spec = kirk_spec(golden)
    p = hjb_oracle.Problem(spec.knots, spec.m, spec.next_terms, spec.cost_terms, spec.dtype)
    N = int(golden["N"])
    J, idx, Js, Is, done = hjb_oracle.sweep(p, N - 1, keep=True)
    assert done == N - 1
    ref, ui = golden["J_star"], golden["u_star_idx"]
    for k in range(N - 1):
        ks = N - 2 - k   # computed k-th (0-based) stage = reference k_s = N-1-k -> plane k_s-1
        assert np.max(np.abs(Js[k] - ref[:, :, ks]) / np.abs(ref[:, :, ks])) < 1e-12
        assert np.array_equal(Is[k], ui[:, :, ks])


def test_c_twin_reproduces_matlab_fixture(orc, golden):
    _abi, c_oracle, hjb_oracle = orc
    spec = kirk_spec(golden)
    N = int(golden["N"])
    out = c_oracle.sweep(_abi, spec, N - 1, keep_J=True, keep_idx=True)
    Js = out["J_stages"].reshape(35, 35, N - 1, order="F")
    Is = out["idx_stages"].reshape(35, 35, N - 1, order="F")
    ref = golden["J_star"][:, :, : N - 1]
    assert np.max(np.abs(Js - ref) / np.abs(ref)) < 1e-12
    assert np.array_equal(Is, golden["u_star_idx"])          # 0-based labels
    assert np.array_equal(out["J"], out["J_stages"][:, 0])    # final = stage k_s = 1


def test_host_class_builds_the_fixture_problem(built, golden):
    """Dynamic_Solver(precision='double') with obj_1.txt's parameters yields the
    fixture's own grid vectors (MATLAB linspace restated bit-exactly)."""
    import hjbdp
    ds = hjbdp.Dynamic_Solver(precision="double")
    ds.N, ds.dx, ds.du = 130, 35, 100
    spec = ds.build_spec()
    assert np.array_equal(spec.knots[0], golden["knots1"]) and np.array_equal(spec.knots[1], golden["knots2"])
    assert np.array_equal(ds._U_mesh, golden["U_mesh"])
    ref = kirk_spec(golden)
    for a in range(2):
        for t, r in zip(spec.next_terms[a], ref.next_terms[a]):
            assert t.dims == r.dims and np.array_equal(t.data, r.data)


def test_float32_tracks_float64(orc, golden):
    _abi, c_oracle, hjb_oracle = orc
    o64 = c_oracle.sweep(_abi, kirk_spec(golden), 40)
    o32 = c_oracle.sweep(_abi, kirk_spec(golden, np.float32), 40)
    assert np.max(np.abs(o32["J"] - o64["J"]) / np.abs(o64["J"])) < 1e-4


CASES = [((9,), (5,), np.float64, False), ((8, 7), (4, 3), np.float64, True), ((7, 6, 5), (3, 2, 2), np.float32, True),
         ((6, 5, 4, 3), (5,), np.float64, False), ((4, 3, 3, 3, 3, 3), (2, 2, 2), np.float64, True)]


@pytest.mark.parametrize("n,m,dtype,nonuniform", CASES)
def test_c_twin_matches_numpy_oracle(orc, n, m, dtype, nonuniform):
    """Same algorithm, different evaluation order (fma vs mul+add): values agree to
    a few ulp; wherever the minimum is not a near-tie the argmin agrees exactly."""
    _abi, c_oracle, hjb_oracle = orc
    spec = random_problem(11, n, m, dtype=dtype, nonuniform=nonuniform)
    term = random_terminal(spec, 3)
    Jc, ic = c_oracle.backup_stage(_abi, spec, term)
    p = hjb_oracle.Problem(spec.knots, spec.m, spec.next_terms, spec.cost_terms, spec.dtype)
    Jn, inp = hjb_oracle.backup_stage(p, term.reshape(n, order="F"))
    Jn = Jn.reshape(-1, order="F")
    inp = inp.reshape(-1, order="F")
    tol = 1e-12 if dtype == np.float64 else 2e-5
    assert np.max(np.abs(Jc - Jn) / np.maximum(1.0, np.abs(Jn))) < tol
    assert np.mean(ic == inp) > 0.98


def test_exact_ties_cascade_order(orc):
    from hjbdp import ProblemSpec, Term
    _abi, c_oracle, hjb_oracle = orc
    k = np.linspace(-1, 1, 5)
    M = np.full((4, 3), 5.0)
    M[0, 1] = M[2, 0] = -1.0   # two joint minimisers: flat label 2 (i1=2,i2=0) and 4 (i1=0,i2=1)
    spec = ProblemSpec([k, k], [4, 3], [[Term((0,), k)], [Term((1,), k)]],
                       [Term((0,), k ** 2), Term((2, 3), M)], dtype=np.float64)
    Jc, ic = c_oracle.backup_stage(_abi, spec, np.zeros(25))
    assert np.all(ic == 4)      # cascade min (dims 9,8,7 of Solver_attitude.m:400-409) prefers the smallest i1 -> label 0 + 4*1
    p = hjb_oracle.Problem(spec.knots, spec.m, spec.next_terms, spec.cost_terms, spec.dtype)
    Jn, inp = hjb_oracle.backup_stage(p, np.zeros((5, 5)))
    assert np.all(inp == 4) and np.array_equal(Jn.reshape(-1, order="F"), Jc)


def test_extrapolation_is_linear(orc):
    """griddedInterpolant 'linear' extrapolates linearly (test/test_griddedInterp.m:20)."""
    from hjbdp import ProblemSpec, Term
    _abi, c_oracle, hjb_oracle = orc
    k = np.linspace(0.0, 1.0, 5)
    shift = np.array([-3.0, 0.0, 2.5])
    spec = ProblemSpec([k], [3], [[Term((0,), k), Term((1,), shift)]], [Term((0,), np.zeros(5))], dtype=np.float64)
    J = 2.0 * k + 1.0          # affine data: interpolation and extrapolation are exact
    Jc, ic = c_oracle.backup_stage(_abi, spec, J)
    assert np.allclose(Jc, 2.0 * (k - 3.0) + 1.0, atol=1e-12) and np.all(ic == 0)


def test_slab_oracle_matches_whole_grid(orc):
    _abi, c_oracle, hjb_oracle = orc
    spec = random_problem(99, (9, 8, 12), (4, 3), dtype=np.float32, spread=0.08)
    term = random_terminal(spec, 3)
    Jw, iw = c_oracle.backup_stage(_abi, spec, term)
    T3 = term.reshape(72, 12, order="F")
    b, e, lo, hi = 4, 8, 2, 2
    Jin = np.asfortranarray(T3[:, b - lo:e + hi]).reshape(-1, order="F")
    Jo, io = c_oracle.backup_stage(_abi, spec, Jin, slab=(b, e, lo, hi))
    assert np.array_equal(Jo.reshape(72, -1, order="F")[:, lo:lo + e - b], Jw.reshape(72, 12, order="F")[:, b:e])
    assert np.array_equal(io, iw.reshape(72, 12, order="F")[:, b:e].reshape(-1, order="F"))


@pytest.mark.parametrize("case", ["random3d", "random4d_f16", "kirk", "rows_of_5", "one_axis", "slab", "tab64_4d", "tab64_slab_f16"])
def test_avx2_twin_equals_scalar_twin(orc, golden, case):
    """The row-vectorised C twin (bench.py's faster CPU baseline) performs the scalar twin's
    operations lane by lane: cost-to-go and argmin bit for bit, with ties, extrapolation, ragged rows and slabs."""
    _abi, c_oracle, hjb_oracle = orc
    slab = None
    if case == "random3d":
        spec = random_problem(5, (19, 8, 12), (4, 3), dtype=np.float32, spread=0.3)
    elif case == "random4d_f16":
        from problems import colsweep_problem
        spec = colsweep_problem(8, (21, 6, 7, 10), j_storage=np.float16)
    elif case == "kirk":
        spec = kirk_spec(golden, dtype=np.float32)
    elif case == "rows_of_5":
        spec = random_problem(6, (5, 4, 3), (7,), dtype=np.float32, spread=0.5, nonuniform=True)
    elif case == "one_axis":
        spec = random_problem(7, (41,), (9,), dtype=np.float32, spread=0.4)
    elif case in ("tab64_4d", "tab64_slab_f16"):
        # the reference's pos-att typing (Solver_pos_att.m:299-327): double query tables, single blend - what bench.py's
        # CPU leg times beside the GPU line
        import hjbdp
        from problems import colsweep_problem
        s64 = (colsweep_problem(31, (21, 6, 7, 10), dtype=np.float64) if case == "tab64_4d"
               else random_problem(32, (13, 7, 12), (5, 2), dtype=np.float64, spread=0.08, nonuniform=True))
        spec = hjbdp.ProblemSpec(s64.knots, s64.m, s64.next_terms, s64.cost_terms, dtype=np.float32, table_dtype=np.float64,
                                 index_base=1, j_storage=np.float16 if case == "tab64_slab_f16" else None)
        if case == "tab64_slab_f16":
            slab = (3, 9, 2, 2)
    else:
        spec = random_problem(99, (9, 8, 12), (4, 3), dtype=np.float32, spread=0.08)
        slab = (4, 8, 2, 2)
    term = random_terminal(spec, 3)
    if case == "kirk":
        term = np.zeros(spec.nS, dtype=np.float32)          # every first-stage total ties between some controls
    if slab is not None:
        inner = spec.nS // spec.n[-1]
        term = np.asfortranarray(term.reshape(inner, -1, order="F")[:, slab[0] - slab[2]:slab[1] + slab[3]]).reshape(-1, order="F")
    Js, i_s = c_oracle.backup_stage(_abi, spec, term, slab=slab)
    Jv, i_v = c_oracle.backup_stage(_abi, spec, term, slab=slab, impl="avx2")
    assert np.array_equal(Js.view(np.uint8), Jv.view(np.uint8)) and np.array_equal(i_s, i_v)
    if case == "random3d":                                   # float64 arithmetic: the vector form says so, never guesses
        spec64 = random_problem(5, (19, 8, 12), (4, 3), dtype=np.float64, spread=0.3)
        with pytest.raises(RuntimeError):
            c_oracle.backup_stage(_abi, spec64, random_terminal(spec64, 3), impl="avx2")


@pytest.mark.parametrize("n,m,nonuniform", [((9, 7, 6), (5,), True), ((6, 5, 4, 5), (3, 2), True), ((11, 8), (7,), False),
                                            ((6, 5, 4), (3, 2, 4), False), ((3, 4, 3, 4, 3), (2, 3, 2), True), ((3, 3, 2, 3, 3, 4), (3, 2, 3), False)])
def test_backup_oracle_against_scipy_interpolant(orc, n, m, nonuniform):
    """The backup itself against an independent implementation of griddedInterpolant's 'linear' semantics on what the
    reference's golden vector does not cover (non-uniform knots, D = 3 .. 6, two and three control dims): scipy's
    RegularGridInterpolator (multilinear, fill_value=None = linear extrapolation) evaluated at the oracle's own next
    states, plus numpy's first-minimum.  float64: 1e-11 relative on J; argmin equal wherever the minimum is not a
    near-tie."""
    from scipy.interpolate import RegularGridInterpolator
    _abi, c_oracle, hjb_oracle = orc
    spec = random_problem(17 + len(n), n, m, dtype=np.float64, nonuniform=nonuniform, spread=0.45)
    term = random_terminal(spec, 9)
    Jc, ic = c_oracle.backup_stage(_abi, spec, term)
    prob = hjb_oracle.Problem(spec.knots, spec.m, [[hjb_oracle.Term(t.dims, t.data) for t in ts] for ts in spec.next_terms],
                              [hjb_oracle.Term(t.dims, t.data) for t in spec.cost_terms], dtype=np.float64)
    full = tuple(n) + tuple(m)
    q = np.stack([np.broadcast_to(prob._sum(prob.next_terms[a]), full).reshape(-1) for a in range(len(n))], axis=1)
    F = RegularGridInterpolator([np.asarray(k, dtype=np.float64) for k in spec.knots], np.asarray(term).reshape(n, order="F"),
                                method="linear", bounds_error=False, fill_value=None)
    tot = (np.broadcast_to(prob._sum(prob.cost_terms), full).reshape(-1) + F(q)).reshape(full)
    tot = tot.reshape(tuple(n) + (-1,))                     # C-order flatten of the control dims: first dim slowest = cascade order
    k = np.argmin(tot, axis=-1)
    Jr = np.take_along_axis(tot, k[..., None], axis=-1)[..., 0].reshape(-1, order="F")
    assert np.max(np.abs(Jc - Jr)) <= 1e-11 * max(1.0, np.max(np.abs(Jr)))
    srt = np.sort(tot, axis=-1)
    clear = ((srt[..., 1] - srt[..., 0]) > 1e-9 * (1 + np.abs(srt[..., 0]))).reshape(-1, order="F")
    sub = np.unravel_index(k.reshape(-1, order="F"), m)     # (i1, ..., iC); the label is column-major, first dim fastest
    lab = np.zeros(k.size, dtype=np.int64)
    mul = 1
    for c in range(len(m)):
        lab += mul * sub[c]
        mul *= m[c]
    assert np.array_equal(ic[clear] - spec.index_base, lab[clear]) and clear.mean() > 0.9


def test_lookup_oracle_against_scipy(orc):
    """The lookup checker itself is checked against an independent implementation
    (scipy RegularGridInterpolator, linear + extrapolation; nearest away from midpoints)."""
    from scipy.interpolate import RegularGridInterpolator
    _abi, c_oracle, hjb_oracle = orc
    rng = np.random.default_rng(5)
    knots = [np.sort(rng.uniform(-1, 1, 9)), np.linspace(-2, 3, 7), np.cumsum(rng.uniform(0.2, 1.0, 5))]
    V = rng.standard_normal((9, 7, 5))
    pts = np.stack([rng.uniform(-1.5, 1.5, 400), rng.uniform(-3, 4, 400), rng.uniform(0, 4, 400)], axis=1)
    lin = c_oracle.lookup(_abi, knots, V, pts, "linear")
    ref = RegularGridInterpolator(knots, V, method="linear", bounds_error=False, fill_value=None)(pts)
    assert np.max(np.abs(lin - ref)) < 1e-11
    inside = pts.copy()
    for a in range(3):
        inside[:, a] = np.clip(inside[:, a], knots[a][0], knots[a][-1])
    near = c_oracle.lookup(_abi, knots, V, inside, "nearest")
    refn = RegularGridInterpolator(knots, V, method="nearest")(inside)
    assert np.mean(near == refn) > 0.99          # only exact-midpoint ties may differ


def test_pos_att_controller_artefact_roundtrip(tmp_path):
    """save_controllers -> MATLAB v5 .mat -> set_controller (Solver_pos_att.m:289, :849-884)."""
    import scipy.io
    import hjbdp
    pa = hjbdp.Solver_pos_att()
    rng = np.random.default_rng(0)
    gv = [np.linspace(-1, 1, 4), np.linspace(-1, 1, 3), np.linspace(-1, 1, 3), np.linspace(-1, 1, 2)]
    ids = rng.integers(1, 10, size=(4, 3, 3, 2))
    from hjbdp.solver_pos_att import vectors_allcomb
    f = vectors_allcomb(pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7)
    pa.controllers["channel_x_controller_1"] = {
        "GridVectors": gv, "F_gI_Values": rng.random((4, 3, 3, 2)).astype(np.float32), "U_Optimal_id": ids,
        "f0_allcomb": f[0], "f1_allcomb": f[1], "f6_allcomb": f[2], "f7_allcomb": f[3]}
    (path,) = pa.save_controllers(tmp_path)
    m = scipy.io.loadmat(path)
    assert m["U_Optimal_id"].dtype == np.float64 and m["F_gI_Values"].dtype == np.float32
    pols = pa.set_controller(path, "x")
    assert pa.Opt_F_Thr0 is pols[0] and pa.Opt_F_Thr7 is pols[3]
    x = (0.9, -0.1, 0.2, 1.0)
    i = (3, 1, 1, 1)
    assert pols[1](*x) == f[1][ids[i] - 1]
    with pytest.raises(ValueError):
        pa.set_controller(path, "q")


def test_oracle_float64_query_mode_against_numpy(orc):
    """hjb_problem.table_dtype = HJB_TAB_F64 in the C twin (Solver_pos_att.m:299-327: double query tables, single
    Values) against an independent numpy evaluation of the same typing: queries summed in float64, located on the
    float64 knots, weight (q - k[c]) / dx in float64 rounded to float32 once, float32 blend (axis 0 first), float32
    cost, first minimum.  Non-uniform knots, 3-D and 4-D, two control dims; also: the mode is not a no-op."""
    import hjbdp
    _abi, orc, _ = orc
    for seed, n, m in ((3, (9, 8, 7), (3, 4)), (4, (7, 6, 5, 6), (5,))):
        s64 = random_problem(seed, n, m, dtype=np.float64, nonuniform=True)
        spec = hjbdp.ProblemSpec(s64.knots, s64.m, s64.next_terms, s64.cost_terms, dtype=np.float32, table_dtype=np.float64)
        term = random_terminal(spec, seed)
        Jc, ic = orc.backup_stage(_abi, spec, term)
        D, G = spec.D, spec.D + spec.C
        full = spec.n + spec.m

        def expand(t, dtype):
            shape = [1] * G
            for ax, d in enumerate(t.dims):
                shape[d] = t.data.shape[ax]
            return np.asarray(t.data, dtype=dtype).reshape(shape)
        J = term.reshape(spec.n, order="F")
        cells, ts = [], []
        for a in range(D):
            q = None
            for t in spec.next_terms[a]:
                e = expand(t, np.float64)
                q = e if q is None else q + e
            q = np.broadcast_to(q, full)
            k = spec.knots[a]
            c = np.clip(np.searchsorted(k, q, side="right") - 1, 0, len(k) - 2)
            w = ((q - k[c]) * (1.0 / (k[c + 1] - k[c]))).astype(np.float32)
            cells.append(c); ts.append(w)
        vals = [J[tuple(cells[a] + ((corner >> a) & 1) for a in range(D))] for corner in range(1 << D)]
        for a in range(D):
            vals = [(vals[j] + (ts[a] * (vals[j + 1] - vals[j])).astype(np.float32)) for j in range(0, len(vals), 2)]
        # the twin's lerp is fmaf(t, v1 - v0, v0): redo the last rounding exactly in float64 (products of two float32
        # are exact in float64, so fma = round(t * d + v0) computed in float64 then rounded once)
        vals = [J[tuple(cells[a] + ((corner >> a) & 1) for a in range(D))] for corner in range(1 << D)]
        for a in range(D):
            nxt = []
            for j in range(0, len(vals), 2):
                d = (vals[j + 1] - vals[j]).astype(np.float32)
                nxt.append((ts[a].astype(np.float64) * d.astype(np.float64) + vals[j].astype(np.float64)).astype(np.float32))
            vals = nxt
        g = None
        for t in spec.cost_terms:
            e = expand(t, np.float32)
            g = e if g is None else (g + e).astype(np.float32)
        tot = (np.broadcast_to(g, full) + vals[0]).astype(np.float32)
        tot = tot.reshape(spec.n + (spec.nU,), order="F") if spec.C == 1 else None
        if tot is None:                                       # two control dims: control dim 0 slowest in the visit order
            t2 = (np.broadcast_to(g, full) + vals[0]).astype(np.float32)
            t2 = np.moveaxis(t2, (D, D + 1), (-2, -1)).reshape(spec.n + (spec.nU,))        # (i1 slow, i2 fast)
            k = np.argmin(t2, axis=-1)
            lab = (k // spec.m[1]) + spec.m[0] * (k % spec.m[1])                          # column-major label
            Jn = np.take_along_axis(t2, k[..., None], axis=-1)[..., 0]
        else:
            k = np.argmin(tot, axis=-1)
            lab = k
            Jn = np.take_along_axis(tot, k[..., None], axis=-1)[..., 0]
        assert np.array_equal(Jn.reshape(-1, order="F"), Jc)
        assert np.array_equal(lab.reshape(-1, order="F").astype(np.int32), ic)
        s32 = hjbdp.ProblemSpec(s64.knots, s64.m, s64.next_terms, s64.cost_terms, dtype=np.float32)
        J32, _ = orc.backup_stage(_abi, s32, term)
        assert not np.array_equal(J32, Jc) and np.allclose(J32, Jc, rtol=1e-4, atol=1e-5)


def test_oracle_single_precision_monitor_sum_is_the_stated_tree(orc):
    """hjb_solve_opts.monitor_single: the oracle's float32 sum follows the order csrc/kernels_reduce.h states (131072
    strided accumulators, pairwise halving per block of 256, 512 block sums folded the same way) - checked against a
    numpy restatement of that order, and against the float64 sum to float32 accuracy."""
    _abi, orc, _ = orc
    spec = random_problem(1, (90, 80, 41), (3,), dtype=np.float32)         # 295200 states: more than one pass of the tree
    out = orc.sweep(_abi, spec, 1, monitor_period=1, monitor_tol=0.0, monitor_single=True)
    out64 = orc.sweep(_abi, spec, 1, monitor_period=1, monitor_tol=0.0)
    J = out["J"]
    NB, NT = 512, 256
    pad = np.zeros(((J.size + NB * NT - 1) // (NB * NT)) * NB * NT, dtype=np.float32)
    pad[:J.size] = J
    acc = np.zeros(NB * NT, dtype=np.float32)
    for row in pad.reshape(-1, NB * NT):
        acc = (acc + row).astype(np.float32)               # adding 0.0f to a float32 is exact: padding changes nothing
    v = acc.reshape(NB, NT).copy()
    s = NT // 2
    while s:
        v[:, :s] = (v[:, :s] + v[:, s:2 * s]).astype(np.float32)
        s //= 2
    part = v[:, 0]
    fin = (part[:NT] + part[NT:]).astype(np.float32)        # block sums t, t + 256 in that order (0 + a + b)
    s = NT // 2
    while s:
        fin[:s] = (fin[:s] + fin[s:2 * s]).astype(np.float32)
        s //= 2
    assert out["last_e"] == float(fin[0])
    assert out64["last_e"] == float(J.astype(np.float64).sum()) or abs(out64["last_e"] - J.astype(np.float64).sum()) < 1e-6
    assert abs(out["last_e"] - out64["last_e"]) <= 1e-5 * abs(out64["last_e"])


def test_oracle_float64_cost_equals_the_materialised_single_of_double_sum(orc):
    """hjb_problem.cost_dtype = HJB_COST_F64: the stage cost as the ordered sum of float64 operands in double, rounded to
    single once, is `J_current_M = single(Qx*x.^2 + Qv*v.^2 + Qw*w.^2 + Qt*t.^2 + (R*f.^2 ...))` of Solver_pos_att.m:800-801.
    On the reference's own pos-att grid the sweep with the five float64 operands ('f64') must equal the sweep with that
    array materialised and passed as ONE float32 term ('exact') bit for bit - scalar twin and AVX2 twin - while the
    float32 operand sum ('terms') differs in the last bits."""
    _abi, c_oracle, hjb_oracle = orc
    import hjbdp
    specs = {}
    for mode in ("exact", "f64", "terms"):
        pa = hjbdp.Solver_pos_att()
        pa.cost_mode = mode
        sx, sv, st, sw = pa.grids()
        specs[mode], _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7,
                                               pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    assert specs["f64"].cost_dtype == np.float64 and specs["f64"].cost_terms[0].data.dtype == np.float64
    assert specs["exact"].cost_dtype is None and len(specs["exact"].cost_terms) == 1
    out = {m: c_oracle.sweep(_abi, s, 12) for m, s in specs.items()}
    assert np.array_equal(out["exact"]["J"].view(np.uint32), out["f64"]["J"].view(np.uint32))
    assert np.array_equal(out["exact"]["idx"], out["f64"]["idx"])
    assert not np.array_equal(out["exact"]["J"], out["terms"]["J"])                   # the float32 operand sum is another function
    assert np.max(np.abs(out["exact"]["J"] - out["terms"]["J"])) <= 1e-5 * np.max(out["exact"]["J"])
    term = random_terminal(specs["f64"], 5)
    Js, i_s = c_oracle.backup_stage(_abi, specs["f64"], term)
    Jv, i_v = c_oracle.backup_stage(_abi, specs["f64"], term, impl="avx2")
    Je, i_e = c_oracle.backup_stage(_abi, specs["exact"], term)
    assert np.array_equal(Js.view(np.uint32), Jv.view(np.uint32)) and np.array_equal(i_s, i_v)
    assert np.array_equal(Js.view(np.uint32), Je.view(np.uint32)) and np.array_equal(i_s, i_e)
