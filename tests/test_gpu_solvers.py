"""GPU parity tests (-m gpu) of the spacecraft solver mirrors (SURVEY 8 rows
a9-a12): the class drives libhjbdp; results are bit-exact against the C oracle
twin on the same problem, plus the invariants SURVEY 8c lists (no reference
artefact exists for these solvers: "parity unpinned")."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import hjbdp
    from hjbdp import _abi
    from oracle import c_oracle
    if hjbdp.device_count() < 1:
        pytest.fail("no HIP device visible")
    return hjbdp, _abi, c_oracle


@pytest.mark.order(7)
def test_solver_position_reference_grid(env):
    """Solver_position.simplified_run on the reference's own 201x201x3 grid, 60 stages."""
    hjbdp, _abi, c_oracle = env
    sp = hjbdp.Solver_position()
    specs = [hjbdp.Solver_position().build_spec(ch) for ch in range(3)]
    sp.simplified_run(n_stages=60)
    assert sp.n_mesh_x == 201 and sp.n_mesh_v == 201      # updated like Solver_position.m:100,104
    for ch in range(3):
        spec, s_x, s_v = specs[ch]
        ref = c_oracle.sweep(_abi, spec, 60)
        assert np.array_equal(sp.F_values[ch].reshape(-1, order="F"), ref["J"])
        assert np.array_equal(sp.U_idx[ch].reshape(-1, order="F"), ref["idx"])
    J = sp.F_values[0]
    assert J.min() >= 0 and J[100, 100] == 0.0                 # J >= 0, J(0,0) = 0
    assert np.allclose(J, J[::-1, ::-1], rtol=1e-9)            # J(x,v) = J(-x,-v) on the symmetric grid
    assert sp.U1_Opt(0.0, 0.0) == 0.0                          # u*(0) = 0
    assert sp.U1_Opt(0.3, 0.2) == -0.26 and sp.U1_Opt(-0.3, -0.2) == 0.26


@pytest.mark.order(7)
def test_solver_position_value_is_monotone_in_horizon(env):
    hjbdp, _abi, c_oracle = env
    sp = hjbdp.Solver_position()
    sp.n_mesh_x = sp.n_mesh_v = 60
    spec, _, _ = sp.build_spec(1)
    with hjbdp.Backup(spec) as bk:
        out = bk.solve(50, keep_J=True)
    Js = out["J_stages"]          # column k_s-1; k_s = 50 computed first
    assert np.all(Js[:, :-1] >= Js[:, 1:] - 1e-12)             # J_k non-decreasing as the horizon grows


@pytest.mark.order(7)
def test_solver_attitude_simplified(env):
    hjbdp, _abi, c_oracle = env
    sa = hjbdp.Solver_attitude(n_mesh_t=70, n_mesh_w_simplified=150)
    sa.simplified_run(n_stages=70)           # (>= 64 stages: the batch's stage loop replays its graph, then runs eagerly)
    assert sa.batch_groups == [3]            # one launch per stage for the three channels
    for ch in range(3):
        spec, s_w, s_t = sa.build_spec_simplified(ch)
        ref = c_oracle.sweep(_abi, spec, 70)
        assert np.array_equal(sa.F_values[ch].reshape(-1, order="F"), ref["J"])
        assert np.array_equal(sa.U_idx[ch].reshape(-1, order="F"), ref["idx"])
    sb = hjbdp.Solver_attitude(n_mesh_t=70, n_mesh_w_simplified=150)
    sb.batch_channels = False                # three chains on threads of their own
    sb.simplified_run(n_stages=70)
    assert sb.batch_groups == [1, 1, 1]
    for ch in range(3):
        assert np.array_equal(sa.F_values[ch], sb.F_values[ch]) and np.array_equal(sa.U_idx[ch], sb.U_idx[ch])


@pytest.mark.order(7)
def test_solver_attitude_relabelled_axes(env):
    """run() hands the library the axes with w3 last (control-nested kernel); bit-exact against the
    oracle on the SAME relabelled problem, and equal to the reference labelling up to lerp-order
    rounding (a few ulp of float32)."""
    hjbdp, _abi, c_oracle = env
    sa = hjbdp.Solver_attitude(n_mesh_w=5, n_mesh_q=4)
    sa.run(n_stages=6)
    assert sa.kernel_variant == 4      # packed kernel (state-dependent inner term on w3)
    spec = sa.build_spec_full()
    pspec, to_old = hjbdp.permute_state_axes(spec, sa.AXIS_ORDER)
    ref = c_oracle.sweep(_abi, pspec, 6)
    assert np.array_equal(sa.F_values.reshape(-1, order="F"), to_old(ref["J"]))
    plain = c_oracle.sweep(_abi, spec, 6)
    assert np.max(np.abs(sa.F_values.reshape(-1, order="F") - plain["J"]) / np.maximum(1e-3, np.abs(plain["J"]))) < 2e-5


@pytest.mark.order(7)
def test_solver_attitude_on_the_fly_model(env):
    """HJB_MODEL_QUAT_EULER321: next angles computed in the stage kernel (variant 4 mode 3).  Bit-exact against
    the oracle's restatement of the same model; equal to the tabulated form up to the few-ulp difference
    between the fixed polynomial atan2/asin and numpy's."""
    hjbdp, _abi, c_oracle = env
    for nw, nq, st in ((5, 4, 5), (7, 6, 3)):
        sa = hjbdp.Solver_attitude(n_mesh_w=nw, n_mesh_q=nq)
        sa.U_vector = np.linspace(-0.11, 0.11, 4)
        mspec = sa.build_spec_model()
        with hjbdp.Backup(mspec) as bk:
            assert bk.info()["kernel_variant"] == 4
            with pytest.raises(hjbdp.HjbError):
                bk.set_option("variant", 0)
            out = bk.solve(st)
        ref = c_oracle.sweep(_abi, mspec, st)
        assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
        rng = np.random.default_rng(nw)
        J0 = (rng.random(mspec.nS) * 3).astype(np.float32)            # rough J: exercises the window fallbacks
        with hjbdp.Backup(mspec) as bk:
            Jg, ig = bk.backup_stage(J0)
        Jr, ir = c_oracle.backup_stage(_abi, mspec, J0)
        assert np.array_equal(Jg, Jr) and np.array_equal(ig, ir)
        pspec, _ = hjbdp.permute_state_axes(sa.build_spec_full(), sa.AXIS_ORDER)
        tab = c_oracle.sweep(_abi, pspec, st)
        assert np.max(np.abs(tab["J"] - ref["J"])) <= 1e-5 * np.max(np.abs(tab["J"]))
        assert np.mean(tab["idx"] == ref["idx"]) > 0.999
    sa.run(n_stages=3, on_the_fly=True)
    assert np.array_equal(np.transpose(sa.F_values, (3, 4, 5, 0, 1, 2)).reshape(-1, order="F"), ref["J"])
    # float16 J storage goes through the same kernel
    hspec = sa.build_spec_model(j_storage=np.float16)
    with hjbdp.Backup(hspec) as bk:
        out = bk.solve(3)
    refh = c_oracle.sweep(_abi, hspec, 3)
    assert np.array_equal(out["J"], refh["J"]) and np.array_equal(out["idx"], refh["idx"])


def _separable_backup_on_device(hjbdp, spec, vecs, timed=False):
    """One backup of `spec` from the separable cost-to-go J(i) = ((vecs[0][i0] + vecs[1][i1]) + ...) built on the device
    (hjb_device_fill_separable: one float32 add per axis, like the checker): J_next, J and the labels live in HBM only
    (library-owned buffers - no torch in this process).  -> (dJ_out, d_idx, seconds or None); caller frees."""
    import time
    isz = spec.idx_np_dtype.itemsize
    dJ = hjbdp.DeviceBuffer(spec.nS * 4)
    dO = hjbdp.DeviceBuffer(spec.nS * 4)
    dI = hjbdp.DeviceBuffer(spec.nS * isz)
    secs = None
    with hjbdp.Backup(spec) as bk:
        info = bk.info()
        assert info["kernel_variant"] == 4 and info["n_states"] == spec.nS
        bk.fill_separable(vecs, dJ)
        bk.check_device_status()                       # synchronises
        t0 = time.perf_counter()
        bk.backup_stage_device(dJ, dO, dI)
        bk.check_device_status()
        if timed:
            secs = time.perf_counter() - t0
    dJ.free()
    return dO, dI, secs


@pytest.mark.order(6)
def test_c3_full_size_51_pow_6(env):
    """BASELINE C3: 51^6 = 1.76e10 states x 11^3 torques, one backup on one GPU.  J_k+1 and J_k are
    70.4 GB each, the argmin labels (uint16: 1331 torque triples) 35 GB: 176 GB of the 288 GB HBM.  J_k+1 is a
    separable sum of per-axis vectors built on the device so that the CPU checker can evaluate any state without
    holding the grid: a sample of states must agree with the oracle bit for bit."""
    hjbdp, _abi, c_oracle = env
    free, total = hjbdp.device_mem_info(0)
    if free < 190 * 2 ** 30:
        pytest.skip("needs 190 GB of free HBM, have %.0f GB" % (free / 2 ** 30))
    sa = hjbdp.Solver_attitude(n_mesh_w=51, n_mesh_q=51)
    sa.U_vector = np.linspace(-0.11, 0.11, 11)
    spec0 = sa.build_spec_model()
    spec = hjbdp.ProblemSpec(spec0.knots, spec0.m, spec0.next_terms, spec0.cost_terms, dtype=np.float32, index_base=spec0.index_base,
                             model=spec0.model, idx_dtype="auto")
    assert spec.nS == 51 ** 6 and spec.nU == 1331 and spec.idx_np_dtype == np.uint16
    rng = np.random.default_rng(51)
    vecs = [(rng.random(n) * (1.0 + a)).astype(np.float32) for a, n in enumerate(spec.n)]
    dO, dI, secs = _separable_backup_on_device(hjbdp, spec, vecs, timed=True)
    line = "C3 51^6 x 11^3, uint16 labels, 176 GB resident: %.2f s per stage, %.3e backups/s" % (secs, spec.nS * spec.nU / secs)
    print(line)
    try:
        from conftest import ROOT
        (ROOT / "gpurun_out").mkdir(exist_ok=True)
        with open(ROOT / "gpurun_out" / "c3_stage_time.log", "a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass
    n = np.array(spec.n, dtype=np.int64)
    sel = rng.integers(0, spec.nS, 300)
    # plus grid corners / edges and the very last state (64-bit indexing)
    corners = [sum(int(c) * int(np.prod(n[:a])) for a, c in enumerate(cs))
               for cs in ((0,) * 6, (50,) * 6, (50, 0, 50, 0, 50, 0), (0, 50, 0, 50, 0, 50), (25,) * 6, (50, 50, 50, 0, 0, 50))]
    sel = np.unique(np.concatenate([sel, np.array(corners, dtype=np.int64), [spec.nS - 1, 2 ** 31 - 1, 2 ** 31, 2 ** 32 + 5]]))
    Jr, ir = c_oracle.backup_states(_abi, spec, vecs, sel)
    try:
        assert np.array_equal(dO.gather(np.float32, sel), Jr)
        assert np.array_equal(dI.gather(np.uint16, sel), ir)
    finally:
        dO.free(); dI.free()


@pytest.mark.extended        # (tests/test_gpu_deep.py::test_6d_24_pow_6_five_stages_deep runs the same grid, kernels and sample five stages deep)
@pytest.mark.order(5)
@pytest.mark.parametrize("form", ["tabulated", "on_the_fly"])
def test_6d_24_pow_6_sampled_states(env, form):
    """The 6-D figure DESIGN.md quotes next to C3: Solver_attitude.run's model on a 24^6 = 1.9e8-state grid x 11^3 torques
    (SURVEY 8d), with the next angles tabulated as the reference does (`Solver_attitude.m:449-504`, K3 mode 2) and
    computed on the fly (mode 3).  One backup from a separable J_k+1; a sample of states - corners, edges, random -
    must agree with the oracle bit for bit."""
    hjbdp, _abi, c_oracle = env
    sa = hjbdp.Solver_attitude(n_mesh_w=24, n_mesh_q=24)
    sa.U_vector = np.linspace(-0.11, 0.11, 11)
    if form == "tabulated":
        spec, _ = hjbdp.permute_state_axes(sa.build_spec_full(), sa.AXIS_ORDER)
    else:
        spec = sa.build_spec_model()
    assert spec.nS == 24 ** 6 and spec.nU == 1331
    rng = np.random.default_rng(24)
    vecs = [(rng.random(n) * (1.0 + a)).astype(np.float32) for a, n in enumerate(spec.n)]
    dO, dI, _ = _separable_backup_on_device(hjbdp, spec, vecs)
    n = np.array(spec.n, dtype=np.int64)
    sel = rng.integers(0, spec.nS, 400)
    corners = [sum(int(c) * int(np.prod(n[:a])) for a, c in enumerate(cs))
               for cs in ((0,) * 6, (23,) * 6, (23, 0, 23, 0, 23, 0), (0, 23, 0, 23, 0, 23), (12,) * 6, (23, 23, 23, 0, 0, 23))]
    sel = np.unique(np.concatenate([sel, np.array(corners, dtype=np.int64), [spec.nS - 1]]))
    Jr, ir = c_oracle.backup_states(_abi, spec, vecs, sel)
    try:
        assert np.array_equal(dO.gather(np.float32, sel), Jr)
        assert np.array_equal(dI.gather(spec.idx_np_dtype, sel), ir)
    finally:
        dO.free(); dI.free()


@pytest.mark.order(5)
def test_6d_24_pow_6_as_slabs_of_the_last_axis(env):
    """C3's kernel mode (K3 mode 3: on-the-fly quaternion model, 64-bit state indexing) in its SLAB form at a real size:
    the attitude model of Solver_attitude.run (attitude-control/Solver_attitude.m:280-287) on 24^6 = 1.9e8 states x 11^3
    torques, swept by hjb_solve_multi as 2 and as 4 slabs of the last axis (w3) - every slab on this box's one GPU, halo
    planes copied device to device per stage - must equal the whole-grid launch on EVERY state, two stages deep (the
    second stage reads what the exchange delivered), and a sample of first-stage states must equal the oracle."""
    _c3_mode_as_slabs(env, 24, (2, 4), 246)


@pytest.mark.extended        # (superseded by the FULL-SIZE rehearsal: tests/test_gpu_deep.py::test_c3_full_size_second_stage_whole_grid_and_as_eight_slabs)
@pytest.mark.order(6)
@pytest.mark.watchdog(900)
def test_c3_mode_32_pow_6_as_eight_slabs(env):
    """north_star: "the state grid shards by outermost axis across up to 8 MI355X".  C3's kernel mode in the 8-GPU form -
    eight slabs of four w3 planes - on a grid five times the 24^6 one: 32^6 = 1.07e9 states x 11^3 torques (J 4.3 GB),
    all slabs on this box's one GPU: equal to the whole-grid launch on every state two stages deep, sampled states equal
    to the oracle (the 8-GPU run of 51^6 itself - 22 GB per GPU - needs hardware the builder does not have)."""
    free, total = env[0].device_mem_info(0)
    if free < 40 * 2 ** 30:
        pytest.skip("needs 40 GB of free HBM")
    _c3_mode_as_slabs(env, 32, (8,), 328)


def _c3_mode_as_slabs(env, n, slab_counts, seed):
    hjbdp, _abi, c_oracle = env
    sa = hjbdp.Solver_attitude(n_mesh_w=n, n_mesh_q=n)
    sa.U_vector = np.linspace(-0.11, 0.11, 11)
    spec0 = sa.build_spec_model()
    spec = hjbdp.ProblemSpec(spec0.knots, spec0.m, spec0.next_terms, spec0.cost_terms, dtype=np.float32, index_base=spec0.index_base,
                             model=spec0.model, idx_dtype="auto")
    assert spec.nS == n ** 6 and spec.nU == 1331 and spec.idx_np_dtype == np.uint16
    rng = np.random.default_rng(seed)
    vecs = [(rng.random(na) * (1.0 + a)).astype(np.float32) for a, na in enumerate(spec.n)]
    term = vecs[0].reshape(-1, 1, 1, 1, 1, 1)
    for a in range(1, 6):                                   # ((v0 + v1) + v2) + ...: one float32 add per axis, like the checker
        shape = [1] * 6
        shape[a] = -1
        term = term + vecs[a].reshape(shape)
    term = np.asfortranarray(term.astype(np.float32)).reshape(-1, order="F")
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["kernel_variant"] == 4
        whole = bk.solve(2, terminal=term, keep_J=True)
    first = whole["J_stages"][:, 1]                         # the stage computed first (k_s = 2)
    sel = np.unique(np.concatenate([rng.integers(0, spec.nS, 300), [0, spec.nS - 1, n ** 5 * (n // 2 - 1) + 7, n ** 5 * (n // 2) - 1]]))
    Jr, ir = c_oracle.backup_states(_abi, spec, vecs, sel)
    assert np.array_equal(first[sel], Jr)
    del first
    for n_slabs in slab_counts:
        with hjbdp.MultiBackup(spec, [0] * n_slabs) as mb:
            infos = [mb.slab_info(i) for i in range(n_slabs)]
            assert all(i["kernel_variant"] == 4 for i in infos) and infos[-1]["end"] == n
            assert [i["end"] - i["begin"] for i in infos] == [n // n_slabs] * n_slabs
            assert any(i["halo_lo"] or i["halo_hi"] for i in infos)
            out = mb.solve(2, terminal=term)
        assert np.array_equal(out["J"], whole["J"]), n_slabs
        assert np.array_equal(out["idx"], whole["idx"]), n_slabs


@pytest.mark.order(7)
def test_solver_attitude_full_6d(env):
    """Solver_attitude.run semantics (6-D x 3-D, single) at a reduced size."""
    hjbdp, _abi, c_oracle = env
    sa = hjbdp.Solver_attitude(n_mesh_w=5, n_mesh_q=4)
    sa.run(n_stages=6, relabel=False)
    spec = sa.build_spec_full()
    ref = c_oracle.sweep(_abi, spec, 6)
    assert np.array_equal(sa.F_values.reshape(-1, order="F"), ref["J"])
    lab = (sa.U_idx[0] - 1) + 3 * (sa.U_idx[1] - 1) + 9 * (sa.U_idx[2] - 1) + 1
    assert np.array_equal(lab.reshape(-1, order="F"), ref["idx"])
    assert set(np.unique(sa.U1_Opt)).issubset({np.float32(-0.11), np.float32(0), np.float32(0.11)})


def _oracle_in_the_mirrors_order(c_oracle, _abi, pa, spec, n_stages, **kw):
    """The oracle's sweep of `spec` in the axis order the mirror runs it (pa.axis_order), mapped back to the reference's."""
    run_spec, to_old = pa._relabel(spec)
    ref = c_oracle.sweep(_abi, run_spec, n_stages, **kw)
    if to_old is not None:
        ref = dict(ref)
        ref["J"], ref["idx"] = to_old(ref["J"]), to_old(ref["idx"])
    return ref


@pytest.mark.order(7)
@pytest.mark.parametrize("form", ["default", "reference"])
def test_solver_pos_att_channel_reference_grid(env, form):
    """One pos-att channel on the reference's 30x30x20x15x9 grid, 12 stages, as the mirror runs it by DEFAULT (the library's
    axis labelling (x, theta, w, v): column-sweep kernel; stage cost = the separable operands summed in double) and in the
    reference's own forms (axis order (x, v, theta, w), the materialised single(double sum) cost table): each bit for bit
    against the oracle on the same problem; the two agree to the rounding of a different lerp order."""
    hjbdp, _abi, c_oracle = env
    pa = hjbdp.Solver_pos_att()
    if form == "reference":
        pa.axis_order, pa.cost_mode = None, "exact"
    else:
        assert pa.axis_order == "auto" and pa.cost_mode == "f64"
    sx, sv, st, sw = pa.grids()
    args = (sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    c = pa.calculate_one_channel_U_Opt(*args, "chx", n_stages=12)
    spec, _ = pa.build_channel_spec(*args)
    ref = _oracle_in_the_mirrors_order(c_oracle, _abi, pa, spec, 12, monitor_period=50, monitor_tol=1e-2)
    assert np.array_equal(c["F_gI_Values"].reshape(-1, order="F"), ref["J"])
    assert np.array_equal(c["U_Optimal_id"].reshape(-1, order="F"), ref["idx"])
    assert c["U_Optimal_id"].min() >= 1 and c["U_Optimal_id"].max() <= 9
    if form == "default":
        with hjbdp.Backup(pa._relabel(spec)[0]) as bk:
            assert bk.info()["kernel_variant"] == 7                  # the default lands on the column-sweep kernel
        pr = hjbdp.Solver_pos_att()
        pr.axis_order, pr.cost_mode = None, "exact"
        r = pr.calculate_one_channel_U_Opt(*args, "chx_ref", n_stages=12)
        assert np.max(np.abs(r["F_gI_Values"] - c["F_gI_Values"])) <= 1e-5 * np.max(np.abs(r["F_gI_Values"]))
        assert np.mean(r["U_Optimal_id"] == c["U_Optimal_id"]) > 0.999


@pytest.mark.order(7)
@pytest.mark.parametrize("grid", ["reference", "small"])
def test_solver_pos_att_all_channels_with_monitor(env, grid):
    """simplified_run (4 channels incl. the thruster-failure one) with the early-stop monitor, as the mirror runs it by default: on the
    reference's own 30x30x20x15 grid the channels land on the column-sweep kernel and those that share a group axis are ONE launch per
    stage (hjb_solve_batch), each with its own monitor sums and stop stage; on a small grid the four land on the table kernel and are
    one launch per stage of that one.  Every channel - values, labels, stop stage - equals the oracle's sweep of that channel, and the same run with the
    channels on threads of their own (batch_channels = False: hjb_solve x 4)."""
    hjbdp, _abi, c_oracle = env
    n_st, period, tol = (69, 10, 290000.0) if grid == "reference" else (120, 10, 5.0)

    def mirror(**kw):
        pa = hjbdp.Solver_pos_att()
        if grid == "small":
            pa.n_mesh_x, pa.n_mesh_v, pa.n_mesh_t, pa.n_mesh_w = 8, 8, 6, 5
        pa.monitor_period, pa.monitor_tol = period, tol
        for k, v in kw.items():
            setattr(pa, k, v)
        return pa
    pa = mirror()
    events = []
    pa.simplified_run(n_stages=n_st, progress=lambda k_s, e, e2, sec: events.append(k_s))
    assert sorted(pa.batch_groups) == ([1, 3] if grid == "reference" else [4]), pa.batch_groups
    assert pa.batched
    assert set(pa.controllers) == {"channel_x_controller_1", "channel_y_controller_1", "channel_z_controller_1",
                                   "channel_x_controller_1_failure"}
    sx, sv, st, sw = pa.grids()
    chan = {"channel_x_controller_1": (st[0], pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2),
            "channel_y_controller_1": (st[1], pa.F_Thr2, pa.F_Thr3, pa.F_Thr8, pa.F_Thr9, pa.Qx2, pa.Qv2, pa.Qt2, pa.Qw2, pa.R2, pa.J3),
            "channel_z_controller_1": (st[2], pa.F_Thr4, pa.F_Thr5, pa.F_Thr10, pa.F_Thr11, pa.Qx3, pa.Qv3, pa.Qt3, pa.Qw3, pa.R3, pa.J1),
            "channel_x_controller_1_failure": (st[0], [0.0], pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)}
    stops = {}
    for name, (s_t, *rest) in chan.items():
        spec, _ = pa.build_channel_spec(sx, sv, s_t, sw, *rest)
        assert pa.monitor_single and spec.table_dtype == np.float64 and spec.idx_np_dtype == np.uint8     # the mirror's defaults: the reference's typing
        ref = _oracle_in_the_mirrors_order(c_oracle, _abi, pa, spec, n_st, monitor_period=period, monitor_tol=tol, monitor_single=True)
        c = pa.controllers[name]
        assert c["stages_done"] == ref["stages_done"] and c["stopped_early"] == ref["stopped_early"], (name, c["stages_done"], ref["stages_done"])
        assert np.array_equal(c["F_gI_Values"].reshape(-1, order="F"), ref["J"]), name
        assert np.array_equal(c["U_Optimal_id"].reshape(-1, order="F"), ref["idx"]), name
        stops[name] = c["stages_done"]
    assert len(pa.controllers["channel_x_controller_1_failure"]["f0_allcomb"]) == 6
    if grid == "reference":
        assert stops["channel_x_controller_1"] == 40 and stops["channel_z_controller_1"] == 50, stops      # x leaves its batch (x, z, failure) one monitor point before the others
    assert events and all(k % period == 0 for k in events)
    pt = mirror(batch_channels=False)
    pt.simplified_run(n_stages=n_st)
    assert not pt.batched
    for name in chan:
        a, b = pa.controllers[name], pt.controllers[name]
        assert a["stages_done"] == b["stages_done"] and np.array_equal(a["F_gI_Values"], b["F_gI_Values"]) and np.array_equal(a["U_Optimal_id"], b["U_Optimal_id"]), name


@pytest.mark.order(7)
def test_dynamic_solver_default_config_runs(env):
    """C1b: the committed constructor defaults (100x100x1000, N=200, single)."""
    hjbdp, _abi, c_oracle = env
    ds = hjbdp.Dynamic_Solver()
    ds.run()
    assert ds.u_star.shape == (100, 100, 200) and ds.u_star.dtype == np.float32
    assert not ds.u_star[:, :, 199].any()
    X, U = ds.get_optimal_path()
    # same qualitative trajectory as the fixture/Kirk Fig. 3-9(b): strong negative first control, decay to the origin
    assert -9.0 < U[0] < -5.5 and np.abs(X[:, -1]).max() < 0.2
    X2, U2 = ds.get_optimal_path(np.array([2.0, 1.0]), "ssu", 1)
    assert ds.ssu_tol == 0.0 and ds.ssu_err_first == 0.0


@pytest.mark.order(3)
def test_c4_full_size_plane_vs_oracle(env):
    """BASELINE configs[3] size (pos-att channel on 120^4 = 2.07e8 cells x 9 thruster combinations,
    non-uniform sym_linspace knots): one stage on the GPU from a smooth terminal cost; one whole
    plane of the last axis is recomputed by the oracle (slab + halos) and must match bit for bit."""
    hjbdp, _abi, c_oracle = env
    pa = hjbdp.Solver_pos_att()
    pa.cost_mode = "terms"
    pa.n_mesh_x = pa.n_mesh_v = pa.n_mesh_t = pa.n_mesh_w = 120
    sx, sv, st, sw = pa.grids()
    spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7,
                                    pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    assert spec.nS == 120 ** 4
    X, V, T = np.meshgrid(sx, sv, st[0], indexing="ij")
    inner = (np.sin(7 * X) + 3 * V ** 2 + np.cos(5 * T)).astype(np.float32).reshape(-1, order="F")
    term = (inner[:, None] * (1.0 + 10.0 * sw[None, :] ** 2).astype(np.float32)).astype(np.float32)   # [120^3, 120]
    with hjbdp.Backup(spec) as bk:
        need = bk.info()
        assert need["kernel_variant"] == 6              # one wave per grid row (long rows, large grid)
        J, idx = bk.backup_stage(term.reshape(-1, order="F"))
    hl, hh = need["halo_needed_lo"], need["halo_needed_hi"]
    assert 6 <= hl <= 9 and 6 <= hh <= 9            # w moves up to ~7.5 cells per stage on this fine grid
    p = 61
    sub = np.asfortranarray(term[:, p - hl:p + 1 + hh]).reshape(-1, order="F")
    Jo, io = c_oracle.backup_stage(_abi, spec, sub, slab=(p, p + 1, hl, hh))
    n3 = 120 ** 3
    assert np.array_equal(Jo.reshape(n3, -1, order="F")[:, hl], J.reshape(n3, 120, order="F")[:, p])
    assert np.array_equal(io, idx.reshape(n3, 120, order="F")[:, p])


@pytest.mark.order(3)
@pytest.mark.parametrize("j_storage", [None, np.float16])
def test_c4_c5_full_size_colsweep_plane_vs_oracle(env, j_storage):
    """BASELINE configs[3] and [4] (C4: pos-att 120^4 x 9 float32; C5: the same with float16 cost-to-go storage) on the
    kernel that is meant for them: the axes relabelled (x, theta, v, w) select the column-sweep stage kernel
    (variant 7) automatically; one stage from a smooth terminal cost, and one whole plane of the last axis is
    recomputed by the oracle (slab + halos) and must match bit for bit."""
    hjbdp, _abi, c_oracle = env
    pa = hjbdp.Solver_pos_att()
    pa.cost_mode = "terms"
    pa.n_mesh_x = pa.n_mesh_v = pa.n_mesh_t = pa.n_mesh_w = 120
    sx, sv, st, sw = pa.grids()
    spec0, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7,
                                     pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    spec, _ = hjbdp.permute_state_axes(spec0, (0, 2, 1, 3))        # (x, theta, v, w): the w-last labelling, wide halo
    if j_storage is not None:
        spec = hjbdp.ProblemSpec(spec.knots, spec.m, spec.next_terms, spec.cost_terms, dtype=np.float32, index_base=1,
                                 j_storage=j_storage, idx_dtype=spec.idx_dtype, table_dtype=spec.table_dtype)
    assert spec.nS == 120 ** 4 and spec.n == (120, 120, 120, 120)
    assert spec.table_dtype == np.float64 and spec.idx_np_dtype == np.uint8      # the mirror's (= the reference's) typing, both storages
    X, T, V = np.meshgrid(sx, st[0], sv, indexing="ij")
    inner = (np.sin(7 * X) + 3 * V ** 2 + np.cos(5 * T)).astype(np.float32).reshape(-1, order="F")
    term = (inner[:, None] * (1.0 + 10.0 * sw[None, :] ** 2).astype(np.float32)).astype(spec.j_dtype)   # [120^3, 120]
    with hjbdp.Backup(spec) as bk:
        need = bk.info()
        assert need["kernel_variant"] == 7
        J, idx = bk.backup_stage(term.reshape(-1, order="F"))
    hl, hh = need["halo_needed_lo"], need["halo_needed_hi"]
    assert 6 <= hl <= 9 and 6 <= hh <= 9            # w moves up to ~7.6 cells per stage on this fine grid
    n3 = 120 ** 3
    for p in (0, 61, 119):                          # both edges (clamped cells) and the interior
        lo, hi = min(hl, p), min(hh, 119 - p)
        sub = np.asfortranarray(term[:, p - lo:p + 1 + hi]).reshape(-1, order="F")
        Jo, io = c_oracle.backup_stage(_abi, spec, sub, slab=(p, p + 1, lo, hi))
        assert np.array_equal(Jo.reshape(n3, -1, order="F")[:, lo], J.reshape(n3, 120, order="F")[:, p]), p
        assert np.array_equal(io, idx.reshape(n3, 120, order="F")[:, p]), p


@pytest.mark.order(4)
def test_c4_bench_order_as_eight_slabs_full_size(env):
    """BASELINE configs[3] as bench.py runs it - pos-att 120^4 x 9, axes (x, theta, w, v), the last axis v sharded - in
    its 8-GPU form on the one GPU of the box: hjb_solve_multi with eight slabs of 15 planes (halo of ONE plane each way,
    interior + boundary strips on their own streams, columns swept in parts) equals the whole-grid sweep bit for bit
    over two stages, and planes of the first stage equal the oracle's (both edges, a slab boundary, the interior)."""
    hjbdp, _abi, c_oracle = env
    pa = hjbdp.Solver_pos_att()
    pa.cost_mode = "terms"
    pa.n_mesh_x = pa.n_mesh_v = pa.n_mesh_t = pa.n_mesh_w = 120
    sx, sv, st, sw = pa.grids()
    spec0, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7,
                                     pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    spec, _ = hjbdp.permute_state_axes(spec0, (0, 2, 3, 1))
    assert spec.n == (120, 120, 120, 120)
    X, T, W = np.meshgrid(sx, st[0], sw, indexing="ij")
    inner = (np.sin(7 * X) + 10.0 * W ** 2 + np.cos(5 * T)).astype(np.float32).reshape(-1, order="F")
    term = (inner[:, None] * (1.0 + 3.0 * sv[None, :] ** 2).astype(np.float32)).astype(np.float32)          # [120^3, 120]
    n3 = 120 ** 3
    with hjbdp.Backup(spec) as bk:
        need = bk.info()
        assert need["kernel_variant"] == 7 and need["halo_needed_lo"] == 1 and need["halo_needed_hi"] == 1
        J1, i1 = bk.backup_stage(term.reshape(-1, order="F"))
        whole = bk.solve(2, terminal=term.reshape(-1, order="F"))
    for p in (0, 14, 15, 64, 119):
        lo, hi = min(1, p), min(1, 119 - p)
        sub = np.asfortranarray(term[:, p - lo:p + 1 + hi]).reshape(-1, order="F")
        Jo, io = c_oracle.backup_stage(_abi, spec, sub, slab=(p, p + 1, lo, hi))
        assert np.array_equal(Jo.reshape(n3, -1, order="F")[:, lo], J1.reshape(n3, 120, order="F")[:, p]), p
        assert np.array_equal(io, i1.reshape(n3, 120, order="F")[:, p]), p
    del J1, i1
    with hjbdp.MultiBackup(spec, [0] * 8) as mb:
        infos = [mb.slab_info(i) for i in range(8)]
        multi = mb.solve(2, terminal=term.reshape(-1, order="F"))
    assert [i["end"] - i["begin"] for i in infos] == [15] * 8 and all(i["split"] for i in infos)
    assert infos[3]["halo_lo"] == 1 and infos[3]["halo_hi"] == 1
    assert np.array_equal(multi["J"], whole["J"]) and np.array_equal(multi["idx"], whole["idx"])


def test_solver_position_closed_loop_rollout(env):
    """Solver_position.get_optimal_path (:189-311): the policies of simplified_run fly the chaser from 1 km behind
    the target towards it: full positive thrust first (x = -1 lies below the grid: 'nearest' clamps to its edge),
    the gap closes monotonically while the thrust is on, the out-of-plane axes stay at rest."""
    hjbdp, _abi, c_oracle = env
    sp = hjbdp.Solver_position()
    sp.simplified_run(n_stages=400)
    T, X, F = sp.get_optimal_path(n_steps=300)
    assert X.shape == (6, 301) and F.shape == (3, 301)
    assert F[0, 0] == pytest.approx(0.26) and np.all(F[0, :100] == pytest.approx(0.26))
    assert np.all(np.diff(X[0, :100]) > 0) and X[0, 300] > X[0, 0]
    assert X[3, 100] == pytest.approx(0.26 * 100 * sp.h, rel=1e-2)           # dv ~ a*t while the thrust is constant
    assert np.max(np.abs(X[2])) < 1e-12 and np.max(np.abs(F[2])) == 0.0      # z stays at the origin


def test_attitude_and_pos_att_closed_loop_rollouts(env):
    """the policies the sweeps leave drive the reference's forward simulators (hjbdp/rollout.py).
    No reference artefact exists for them; pinned by invariants: unit quaternions, thrusters only at their two
    levels, the attitude error and rates decay under the simplified attitude policies, the pos-att chaser moves
    toward the target along x while its attitude error shrinks."""
    hjbdp, _abi, c_oracle = env
    sa = hjbdp.Solver_attitude(n_mesh_t=120, n_mesh_w_simplified=200)
    sa.simplified_run(n_stages=600)
    T, X, U = sa.get_optimal_path_simplified_testode45(n_steps=500)
    assert X.shape == (501, 7) and np.allclose(np.linalg.norm(X[:, 3:], axis=1), 1.0, atol=1e-3)
    assert set(np.unique(U)).issubset({-0.11, 0.0, 0.11}) and np.any(U != 0)
    err0, err1 = np.linalg.norm(X[0, 3:6]), np.linalg.norm(X[500, 3:6])
    assert err1 < err0                                         # 2.5 s of control already reduce the attitude error
    # the 6-D policy of run() at a small size: one lookup per stage in the six-dimensional tables
    sa6 = hjbdp.Solver_attitude(n_mesh_w=7, n_mesh_q=6)
    sa6.run(n_stages=8)
    X6, U6, XA = sa6.get_optimal_path(n_steps=50)
    assert X6.shape == (7, 51) and np.allclose(np.linalg.norm(X6[3:, :], axis=0), 1.0, atol=1e-12)
    assert set(np.unique(U6)).issubset({np.float32(-0.11), np.float32(0.0), np.float32(0.11)} | {-0.11, 0.0, 0.11})
    # pos-att: three channels, 13-state closed loop
    pa = hjbdp.Solver_pos_att()
    pa.cost_mode = "terms"
    pa.simplified_run(n_stages=300)
    T, Xp, F, FM = pa.get_optimal_path(n_steps=200)
    assert Xp.shape == (201, 13) and np.allclose(np.linalg.norm(Xp[:, 6:10], axis=1), 1.0, atol=1e-3)
    assert set(np.unique(np.abs(F))).issubset({0.0, 0.13}) and np.any(F != 0)
    assert Xp[200, 0] > Xp[0, 0]                               # from 100 m behind (-0.1 km) toward the target
    assert np.all(np.abs(FM[:200, 0:3]) <= 2 * 0.13 / pa.Mass + 1e-12) and np.all(np.abs(FM[:200, 3:6]) <= 2 * 0.13 * pa.T_dist + 1e-12)


@pytest.mark.order(8)
@pytest.mark.watchdog(300)
def test_solve_batch_randomised_stress_slice(env):
    """Ten seconds of tools/stress_batch.py: random batches (column-sweep problems of one group axis, small problems on the table kernel),
    stage counts around the 32-stage graph, monitors that stop problems at different stages - hjb_solve_batch equals hjb_solve per
    problem in values, labels, stages done and stop flags.  (The long form: profiles/r06_stress_batch.txt.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_batch.py"), "10", "11"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "stress ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
