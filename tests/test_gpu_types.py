"""GPU parity tests (-m gpu) of the typing options of the C ABI, through the C ABI, against the oracle:

* hjb_problem.idx_dtype    - argmin labels stored as uint8 / uint16 (MATLAB's U_Optimal_id of
                             pos-att/Solver_pos_att.m:272 holds 9 distinct values) in every stage kernel,
* hjb_problem.table_dtype  - HJB_TAB_F64: the reference's pos-att typing (Solver_pos_att.m:299-327: double query
                             tables, single F_gI.Values): (cell, weight) built in float64, weight rounded once,
* hjb_solve_opts.monitor_single - the early-stop monitor's sum of J accumulated in float32 (Solver_pos_att.m:273-285),
* the device-buffer helpers (hjb_device_malloc / copy / fill_separable / gather).

Bars: J and argmin labels BIT-EXACT against the C twin (oracle/hjb_oracle.c)."""
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


def _respec(hjbdp, spec, **kw):
    args = dict(dtype=spec.dtype, index_base=spec.index_base, j_storage=None if spec.j_dtype == spec.dtype else spec.j_dtype,
                idx_dtype=spec.idx_dtype, table_dtype=spec.table_dtype)
    args.update(kw)
    return hjbdp.ProblemSpec(spec.knots, spec.m, spec.next_terms, spec.cost_terms, **args)


def _cases():
    from problems import colsweep_problem, nested_problem, random_problem
    return [
        # (name, spec, variants to force)
        ("generic3", random_problem(5, (9, 8, 7), (4, 5), dtype=np.float32, index_base=1), (0, 5)),
        ("generic2_f64", random_problem(6, (12, 10), (30,), dtype=np.float64), (0, 3, 5)),
        ("nested3", nested_problem(8, (9, 8, 10), (3, 4, 6), dtype=np.float32), (1, 2, 4, 5)),
        ("nested2_f64", nested_problem(9, (11, 9), (7,), dtype=np.float64), (1, 5)),
        ("colsweep", colsweep_problem(720, (70, 9, 8, 11), nU=9), (7, 6, 5)),
        ("colsweep_f16", colsweep_problem(721, (66, 8, 9, 10), nU=9, j_storage=np.float16), (7, 6, 5)),
    ]


@pytest.mark.parametrize("idx_dtype", [np.uint8, np.uint16, "auto"])
def test_narrow_argmin_labels_every_variant(env, idx_dtype):
    """Every stage kernel writes its labels in the width the problem asks for; values equal the oracle's (int32)."""
    hjbdp, _abi, c_oracle = env
    from problems import random_terminal
    for name, spec0, variants in _cases():
        if idx_dtype == np.uint8 and spec0.nU - 1 + spec0.index_base > 255:
            continue
        spec = _respec(hjbdp, spec0, idx_dtype=idx_dtype)
        term = random_terminal(spec0, 4).astype(spec.j_dtype)
        ref = c_oracle.sweep(_abi, spec, 3, terminal=term, keep_idx=True)
        for v in variants:
            with hjbdp.Backup(spec, variant=v) as bk:
                info = bk.info()
                assert info["kernel_variant"] == v
                assert info["idx_bytes"] == spec.idx_np_dtype.itemsize, (name, v)
                out = bk.solve(3, terminal=term, keep_idx=True, monitor_period=1, monitor_tol=0.0)
                assert out["idx"].dtype == spec.idx_np_dtype and out["idx_stages"].dtype == spec.idx_np_dtype
                assert np.array_equal(out["J"], ref["J"]), (name, v)
                assert np.array_equal(out["idx_stages"], ref["idx_stages"]), (name, v)
                assert out["last_e2"] == float(ref["idx_stages"][:, 0].astype(np.float64).sum()
                                               - ref["idx_stages"][:, 1].astype(np.float64).sum())   # monitor reads narrow labels
                Jo, io = bk.backup_stage(term)
                assert io.dtype == spec.idx_np_dtype
                assert np.array_equal(io, ref["idx_stages"][:, 2]), (name, v)


def test_label_width_is_validated(env):
    hjbdp, _abi, c_oracle = env
    from problems import random_problem
    spec = _respec(hjbdp, random_problem(3, (6, 5), (300,), dtype=np.float32, index_base=1), idx_dtype=np.uint8)
    with pytest.raises(hjbdp.HjbError) as e:
        hjbdp.Backup(spec)
    assert e.value.status == _abi.HJB_E_INVALID and "cannot hold" in str(e.value)
    auto = _respec(hjbdp, spec, idx_dtype="auto")
    with hjbdp.Backup(auto) as bk:
        assert bk.info()["idx_bytes"] == 2
    with hjbdp.Backup(_respec(hjbdp, random_problem(3, (6, 5), (255,), dtype=np.float32, index_base=1), idx_dtype="auto")) as bk:
        assert bk.info()["idx_bytes"] == 1          # labels 1..255
    with hjbdp.Backup(_respec(hjbdp, random_problem(3, (6, 5), (256,), dtype=np.float32, index_base=1), idx_dtype="auto")) as bk:
        assert bk.info()["idx_bytes"] == 2          # label 256 needs the second byte


def test_temporal_blocking_and_multi_with_narrow_labels(env):
    """K9 (several stages per launch) and the in-library multi-slab sweep carry the label width too."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, random_terminal
    sp = hjbdp.Solver_position()
    spec0, _, _ = sp.build_spec(0)
    spec = _respec(hjbdp, spec0, idx_dtype="auto")
    ref = c_oracle.sweep(_abi, spec, 40)
    with hjbdp.Backup(spec) as bk:
        bk.set_option("temporal", 2)
        out = bk.solve(40)
    assert out["idx"].dtype == np.uint8
    assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
    cs = _respec(hjbdp, colsweep_problem(33, (40, 7, 6, 12), nU=9, gax=2), idx_dtype=np.uint8)
    term = random_terminal(cs, 2)
    ref = c_oracle.sweep(_abi, cs, 5, terminal=term, keep_idx=True, monitor_period=2, monitor_tol=0.0)
    with hjbdp.MultiBackup(cs, [0, 0, 0]) as mb:
        out = mb.solve(5, terminal=term, keep_idx=True, monitor_period=2, monitor_tol=0.0)
    assert out["idx"].dtype == np.uint8 and out["idx_stages"].dtype == np.uint8
    assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx_stages"], ref["idx_stages"])
    assert out["last_e2"] == ref["last_e2"] and out["last_e"] == ref["last_e"]


def _pos_att_spec(hjbdp, n, table_dtype, **kw):
    pa = hjbdp.Solver_pos_att()
    pa.n_mesh_x, pa.n_mesh_v, pa.n_mesh_t, pa.n_mesh_w = n
    pa.table_dtype = table_dtype
    pa.cost_mode = "terms"          # separable cost operands: what the column-sweep kernel takes
    sx, sv, st, sw = pa.grids()
    spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7,
                                    pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    return _respec(hjbdp, spec, **kw) if kw else spec


@pytest.mark.parametrize("j_storage", [None, np.float16])
def test_double_query_tables_pos_att(env, j_storage):
    """The reference's pos-att typing on the pos-att channel itself (reference axis order and the relabelled one):
    kernels 5, 6, 7 equal the oracle's float64-query mode bit for bit; the float32-query mode differs from it (the
    option is not a no-op); kernels that evaluate terms in float32 are refused."""
    hjbdp, _abi, c_oracle = env
    from problems import random_terminal
    spec = _pos_att_spec(hjbdp, (70, 6, 5, 10), np.float64, j_storage=j_storage, idx_dtype="auto")
    assert spec.next_terms[0][0].data.dtype == np.float64 and spec.cost_terms[0].data.dtype == np.float32
    term = random_terminal(spec, 3).astype(spec.j_dtype)
    ref = c_oracle.sweep(_abi, spec, 5, terminal=term, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["table_dtype"] == _abi.HJB_TAB_F64 and bk.info()["kernel_variant"] in (5, 6, 7)
        for v in (0, 1, 2, 3, 4):
            with pytest.raises(hjbdp.HjbError) as e:
                bk.set_option("variant", v)
            assert e.value.status == _abi.HJB_E_UNSUPPORTED
        for v in (5, 6):
            bk.set_option("variant", v)
            out = bk.solve(5, terminal=term, keep_J=True, keep_idx=True)
            assert np.array_equal(out["J_stages"], ref["J_stages"]), v
            assert np.array_equal(out["idx_stages"], ref["idx_stages"]), v
    order = hjbdp.suggest_axis_order(spec)
    assert order == (0, 2, 3, 1)
    fast, to_old = hjbdp.permute_state_axes(spec, order)
    assert fast.table_dtype == np.float64
    tf = np.transpose(term.reshape(spec.n, order="F"), order).reshape(-1, order="F")
    reff = c_oracle.sweep(_abi, fast, 5, terminal=tf, keep_J=True, keep_idx=True)
    with hjbdp.Backup(fast, variant=7) as bk:
        assert bk.info()["kernel_variant"] == 7 and bk.info()["table_dtype"] == _abi.HJB_TAB_F64
        out = bk.solve(5, terminal=tf, keep_J=True, keep_idx=True)
    assert np.array_equal(out["J_stages"], reff["J_stages"]) and np.array_equal(out["idx_stages"], reff["idx_stages"])
    # ... and the option changes results: the float32-query sweep of the same channel is a different (close) function
    s32 = _pos_att_spec(hjbdp, (70, 6, 5, 10), None, j_storage=j_storage, idx_dtype="auto")
    r32 = c_oracle.sweep(_abi, s32, 5, terminal=term)
    a, b = ref["J"].astype(np.float64), r32["J"].astype(np.float64)
    assert not np.array_equal(a, b)
    assert np.max(np.abs(a - b)) <= (2e-3 if j_storage else 1e-4) * np.max(np.abs(a))


def test_double_query_tables_generic_shapes_and_slabs(env):
    """HJB_TAB_F64 on synthetic shapes (2-D .. 4-D, non-uniform knots, two control dims), whole grids and a slab."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, random_problem, random_terminal
    for seed, n, m, nonuni in ((1, (14, 11), (5,), False), (2, (9, 8, 7), (3, 4), True), (3, (7, 6, 5, 6), (6,), True)):
        spec = _respec(hjbdp, random_problem(seed, n, m, dtype=np.float64, nonuniform=nonuni, index_base=1), dtype=np.float32,
                       table_dtype=np.float64)
        term = random_terminal(spec, seed)
        ref = c_oracle.sweep(_abi, spec, 3, terminal=term)
        with hjbdp.Backup(spec) as bk:
            assert bk.info()["kernel_variant"] >= 5
            out = bk.solve(3, terminal=term)
        assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"]), n
    cs0 = colsweep_problem(44, (66, 9, 8, 12), nU=9, dtype=np.float64)
    cs = _respec(hjbdp, cs0, dtype=np.float32, table_dtype=np.float64)
    term = random_terminal(cs, 9)
    ref = c_oracle.sweep(_abi, cs, 4, terminal=term)
    with hjbdp.Backup(cs, variant=7) as bk:
        out = bk.solve(4, terminal=term)
    assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
    with hjbdp.MultiBackup(cs, [0, 0, 0]) as mb:                  # slabs: the float64 build runs per slab handle
        outm = mb.solve(4, terminal=term)
    assert np.array_equal(outm["J"], ref["J"]) and np.array_equal(outm["idx"], ref["idx"])


def test_monitor_single_precision_sum(env):
    """hjb_solve_opts.monitor_single: the sum of J in float32 in the library's stated order = the oracle's restatement of
    that order, so the early stop lands on the same stage; the float64 monitor on the same sweep sees different deltas."""
    hjbdp, _abi, c_oracle = env
    spec = _pos_att_spec(hjbdp, (30, 30, 20, 15), np.float64, idx_dtype="auto")
    kw = dict(monitor_period=10, monitor_tol=1e-2)
    ref = c_oracle.sweep(_abi, spec, 60, monitor_single=True, **kw)
    ref64 = c_oracle.sweep(_abi, spec, 60, **kw)
    with hjbdp.Backup(spec) as bk:
        out = bk.solve(60, monitor_single=True, **kw)
        out64 = bk.solve(60, **kw)
        bk.set_option("monitor_single", 1)                # the flat API's way of asking for it
        out_opt = bk.solve(60, **kw)
    for o, r in ((out, ref), (out64, ref64), (out_opt, ref)):
        assert o["stages_done"] == r["stages_done"] and o["stopped_early"] == r["stopped_early"]
        assert o["last_e"] == r["last_e"] and o["last_e2"] == r["last_e2"]
        assert np.array_equal(o["J"], r["J"]) and np.array_equal(o["idx"], r["idx"])
    assert out["last_e"] != out64["last_e"]               # the two monitors do not see the same numbers
    with hjbdp.MultiBackup(spec, [0, 0]) as mb:
        o = _abi.hjb_solve_opts()
        o.n_stages, o.monitor_period, o.monitor_tol, o.monitor_single = 4, 2, 0.0, 1
        st = mb.lib.hjb_solve_multi(mb._m, o, None)
        assert st == _abi.HJB_E_UNSUPPORTED


def test_device_buffer_helpers(env):
    """hjb_device_malloc / copy / fill_separable / gather + hjb_backup_stage_device on library-owned buffers: what a
    host without a HIP binding of its own uses (and what the 51^6 / 24^6 tests below run on)."""
    hjbdp, _abi, c_oracle = env
    from problems import random_problem, random_terminal
    spec = _respec(hjbdp, random_problem(77, (12, 10, 9), (4, 4), dtype=np.float32), idx_dtype="auto")
    free, total = hjbdp.device_mem_info(0)
    assert 0 < free <= total and total > 100 * 2 ** 30
    rng = np.random.default_rng(2)
    vecs = [rng.random(n).astype(np.float32) for n in spec.n]
    Jsep = ((vecs[0][:, None, None] + vecs[1][None, :, None]) + vecs[2][None, None, :]).reshape(-1, order="F")
    with hjbdp.DeviceBuffer(spec.nS * 4) as dJ, hjbdp.DeviceBuffer(spec.nS * 4) as dO, hjbdp.DeviceBuffer(spec.nS) as dI, \
            hjbdp.Backup(spec) as bk:
        bk.fill_separable(vecs, dJ)
        assert np.array_equal(dJ.download(np.float32), Jsep)
        bk.backup_stage_device(dJ, dO, dI)
        bk.check_device_status()
        Jr, ir = c_oracle.backup_stage(_abi, spec, Jsep)
        assert np.array_equal(dO.download(np.float32), Jr) and np.array_equal(dI.download(np.uint8), ir)
        sel = rng.integers(0, spec.nS, 500)
        assert np.array_equal(dO.gather(np.float32, sel), Jr[sel]) and np.array_equal(dI.gather(np.uint8, sel), ir[sel])
        term = random_terminal(spec, 1)
        dJ.upload(term)
        assert np.array_equal(dJ.download(np.float32), term)


def test_tab64_table_build_failure_is_loud(env, monkeypatch):
    """A float64-query handle whose (cell, t) table build fails (here: the float64 scratch allocation, forced by the
    library's test hook) is NOT handed out on the float32 copies of its terms: hjb_create fails with the build's status
    and a message that names the remedy.  (it used to fall back to the generic kernel silently.)"""
    hjbdp, _abi, c_oracle = env
    spec = _pos_att_spec(hjbdp, (30, 12, 10, 9), np.float64, idx_dtype="auto")
    lib = hjbdp.load_library()
    monkeypatch.setenv("HJBDP_TEST_FAIL_TAB64_SCRATCH", "1")       # the environment does nothing
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["table_dtype"] == _abi.HJB_TAB_F64
    monkeypatch.delenv("HJBDP_TEST_FAIL_TAB64_SCRATCH")
    assert lib.hjb_test_hook(b"no_such_key", 1) == _abi.HJB_E_INVALID
    assert lib.hjb_test_hook(b"fail_tab64_scratch", 1) == _abi.HJB_OK
    try:
        with pytest.raises(hjbdp.HjbError) as ei:
            hjbdp.Backup(spec)
        assert ei.value.status == _abi.HJB_E_NOMEM and "scratch" in str(ei.value)
        # float32 queries are not affected by the hook, and without it the float64 build works
        with hjbdp.Backup(_pos_att_spec(hjbdp, (30, 12, 10, 9), None, idx_dtype="auto")) as bk:
            assert bk.info()["table_dtype"] == _abi.HJB_TAB_DEFAULT
    finally:
        assert lib.hjb_test_hook(b"fail_tab64_scratch", 0) == _abi.HJB_OK
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["table_dtype"] == _abi.HJB_TAB_F64 and bk.info()["kernel_variant"] >= 5


def test_cost64_table_build_failure_is_loud(env):
    """a HJB_COST_F64 handle whose tables cannot be built is not handed out on variant 0 (which would fail
    every later launch with a message about kernels 5 and 7): hjb_create fails with the build's status."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem
    s0 = colsweep_problem(77, (70, 9, 8, 11), nU=9, gax=3, cost="fast", dtype=np.float64)
    spec = hjbdp.ProblemSpec(s0.knots, s0.m, s0.next_terms, s0.cost_terms, dtype=np.float32, index_base=1, idx_dtype="auto",
                             cost_dtype=np.float64)
    s32 = hjbdp.ProblemSpec(s0.knots, s0.m, s0.next_terms, s0.cost_terms, dtype=np.float32, index_base=1, idx_dtype="auto")
    lib = hjbdp.load_library()
    assert lib.hjb_test_hook(b"fail_tabled_alloc", 1) == _abi.HJB_OK
    try:
        with pytest.raises(hjbdp.HjbError) as ei:
            hjbdp.Backup(spec)
        assert ei.value.status == _abi.HJB_E_NOMEM and "table" in str(ei.value)
        with hjbdp.Backup(s32, variant=0) as bk:            # a float32-cost problem still gets the general kernel
            assert bk.info()["kernel_variant"] == 0
    finally:
        assert lib.hjb_test_hook(b"fail_tabled_alloc", 0) == _abi.HJB_OK
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["cost_dtype"] == _abi.HJB_COST_F64 and bk.info()["kernel_variant"] in (5, 7)


@pytest.mark.parametrize("j_storage", [None, np.float16])
def test_float64_cost_terms_every_serving_kernel(env, j_storage):
    """hjb_problem.cost_dtype = HJB_COST_F64: cost terms float64, the stage cost summed in double and rounded once
    (Solver_pos_att.m:800-801).  The tabled kernel (5) and the column sweep (7, one-load and two-load forms, both group
    axes) against the oracle bit for bit, with float16 storage, irrational cost values (the double sum really differs from
    the float32 one), slabs; kernels that sum in float32 are refused; on the reference's pos-att grid the 'f64' sweep equals
    the 'exact' sweep (the materialised single(double sum)) bit for bit, monitor included."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, random_problem, random_terminal
    rng = np.random.default_rng(64)
    for seed, gax, cost in ((1, 3, "fast"), (2, 2, "fast"), (3, 3, "step01")):
        s0 = colsweep_problem(900 + seed, (70, 9, 8, 11), nU=9, gax=gax, cost=cost, dtype=np.float64)
        ct = [hjbdp.Term(t.dims, np.asarray(t.data) * (1.0 + 1e-3 * rng.random(np.asarray(t.data).shape))) for t in s0.cost_terms]
        spec = hjbdp.ProblemSpec(s0.knots, s0.m, s0.next_terms, ct, dtype=np.float32, index_base=1, idx_dtype="auto",
                                 cost_dtype=np.float64, j_storage=j_storage)
        s32 = hjbdp.ProblemSpec(s0.knots, s0.m, s0.next_terms, ct, dtype=np.float32, index_base=1, idx_dtype="auto", j_storage=j_storage)
        term = random_terminal(spec, seed).astype(spec.j_dtype)
        ref = c_oracle.sweep(_abi, spec, 4, terminal=term)
        assert not np.array_equal(ref["J"], c_oracle.sweep(_abi, s32, 4, terminal=term)["J"])      # the typing matters
        for variant, dpp in ((5, 1), (7, 1), (7, 0)):
            with hjbdp.Backup(spec, variant=variant) as bk:
                if variant == 7:
                    bk.set_option("cs_dpp", dpp)
                inf = bk.info()
                assert inf["kernel_variant"] == variant and inf["cost_dtype"] == _abi.HJB_COST_F64
                out = bk.solve(4, terminal=term)
            assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"]), (seed, variant, dpp)
        with hjbdp.Backup(spec) as bk:
            assert bk.info()["kernel_variant"] in (5, 7)
            for v in (0, 6):
                with pytest.raises(hjbdp.HjbError):
                    bk.set_option("variant", v)
        with hjbdp.MultiBackup(spec, [0, 0, 0]) as mb:
            outm = mb.solve(4, terminal=term)
        assert np.array_equal(outm["J"], ref["J"]) and np.array_equal(outm["idx"], ref["idx"])
    # two control dims, 3-D, through the tabled kernel
    g0 = random_problem(71, (9, 8, 7), (4, 5), dtype=np.float64, nonuniform=True, index_base=1)
    gen = hjbdp.ProblemSpec(g0.knots, g0.m, g0.next_terms, g0.cost_terms, dtype=np.float32, index_base=1, cost_dtype=np.float64,
                            j_storage=j_storage)
    term = random_terminal(gen, 7).astype(gen.j_dtype)
    ref = c_oracle.sweep(_abi, gen, 3, terminal=term)
    with hjbdp.Backup(gen) as bk:
        assert bk.info()["kernel_variant"] == 5
        out = bk.solve(3, terminal=term)
    assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
    if j_storage is None:
        outs = {}
        for mode in ("exact", "f64"):
            pa = hjbdp.Solver_pos_att()
            pa.cost_mode = mode
            sx, sv, st, sw = pa.grids()
            outs[mode] = pa.calculate_one_channel_U_Opt(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1,
                                                        pa.Qt1, pa.Qw1, pa.R1, pa.J2, "c_" + mode, n_stages=150)
        a, b = outs["exact"], outs["f64"]
        assert np.array_equal(a["F_gI_Values"].view(np.uint32), b["F_gI_Values"].view(np.uint32))
        assert np.array_equal(a["U_Optimal_id"], b["U_Optimal_id"]) and a["stages_done"] == b["stages_done"]


@pytest.mark.parametrize("cost_mode", ["f64", "terms"])
def test_reference_grid_on_the_fast_axes_every_part_count(env, cost_mode):
    """The reference's own 30 x 30 x 20 x 15 x 9 pos-att channel swept on relabelled axes (x, theta, v, w) by the column-sweep kernel,
    as `Solver_pos_att.axis_order = "auto"` runs it, with a column cut into 1 .. 20 parts (round 5's automatic choice: ten, two steps
    each) and the stage cost summed in double ('f64') or in single ('terms'): bit-exact against the oracle on the same relabelled
    problem.  Regression: the float64-cost instantiation crashed with eight or more parts - the hand-written batch of scalar loads in
    the kernel's set-up lacked early-clobber outputs, and a destination took the address registers of a later load of the batch."""
    hjbdp, _abi, c_oracle = env
    pa = hjbdp.Solver_pos_att()
    pa.cost_mode = cost_mode
    sx, sv, st, sw = pa.grids()
    spec0, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    order = hjbdp.suggest_axis_order(spec0)
    assert order is not None and tuple(order)[0] == 0
    spec, _ = hjbdp.permute_state_axes(spec0, order)
    ref = c_oracle.sweep(_abi, spec, 6)
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["kernel_variant"] == 7
        auto = bk.get_option("cs_split")
        assert auto >= 8                                        # short columns on a launch below one round of the wave slots
        for parts in (0, 1, 3, 8, 10, 16, 20):
            bk.set_option("cs_split", parts)
            out = bk.solve(6)
            assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"]), parts
    # the mirror on the fast axes: the same labels as in the reference's axis order but for a few in 10^4 (a different lerp order)
    pa2 = hjbdp.Solver_pos_att()
    pa2.cost_mode, pa2.axis_order = cost_mode, "auto"
    a = pa2.calculate_one_channel_U_Opt(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2, "fast", n_stages=60)
    b = pa.calculate_one_channel_U_Opt(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2, "ref", n_stages=60)
    assert a["F_gI_Values"].shape == b["F_gI_Values"].shape == (30, 30, 20, 15)
    assert np.max(np.abs(a["F_gI_Values"] - b["F_gI_Values"])) <= 1e-4 * np.max(np.abs(b["F_gI_Values"]))
    assert np.mean(a["U_Optimal_id"] == b["U_Optimal_id"]) > 0.999
