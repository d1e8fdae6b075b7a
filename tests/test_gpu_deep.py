"""GPU parity tests (-m gpu), DEEP in the stage direction at the BASELINE sizes and stage counts.

The other full-size tests check one or two stages from a smooth terminal cost.  These run the sweeps the reference runs -
C2 100 stages (BASELINE configs[1]), C4 / C5 200 stages with the early-stop monitor at period 50 in the bench's exact
typing (pos-att/Solver_pos_att.m:270-286), C1b 100x100x1000 over its 199 stages (test/Dynamic_Solver.m:86-102),
Solver_attitude.simplified_run's 1000x300 grid over 200 stages (attitude-control/Solver_attitude.m:236-247) - through
hjb_solve (hipGraph replay, ping-pong buffers, write-out wrap-around, monitor reduction at 2e8 cells), and have the oracle
recompute stage k from the GPU's OWN stage k+1: a whole plane of the last axis for the big grids, the whole grid for the
small ones.  Bars: J and labels BIT-EXACT; the monitor's sums, differences and stop decision equal the oracle's on the
same J (single-precision sum, difference and comparison: Solver_pos_att.m:274-282)."""
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


def _plane_from_previous(c_oracle, _abi, spec, J_prev, J_cur, idx_cur, planes, hl, hh):
    """Stage k's planes `planes` of the last axis recomputed by the oracle from the GPU's own stage k+1 (slab + halos)."""
    nl = spec.n[-1]
    inner = spec.nS // nl
    Jp = J_prev.reshape(inner, nl, order="F")
    Jc = J_cur.reshape(inner, nl, order="F")
    ic = idx_cur.reshape(inner, nl, order="F")
    for p in planes:
        lo, hi = min(hl, p), min(hh, nl - 1 - p)
        sub = np.asfortranarray(Jp[:, p - lo:p + 1 + hi]).reshape(-1, order="F")
        Jo, io = c_oracle.backup_stage(_abi, spec, sub, slab=(p, p + 1, lo, hi))
        assert np.array_equal(Jo.reshape(inner, -1, order="F")[:, lo], Jc[:, p]), ("J", p)
        assert np.array_equal(io, ic[:, p]), ("labels", p)


@pytest.mark.order(2)
@pytest.mark.watchdog(600)
def test_c2_100_stages_deep(env):
    """BASELINE configs[1]: 101^3 x 21^3, 100 stages, uint16 labels (the bench's typing), every stage kept; the oracle
    recomputes one plane of stages 100 (from the zero terminal cost), 99, 51, 50, 2 and 1 from the GPU's previous stage."""
    hjbdp, _abi, c_oracle = env
    import bench
    spec, _ = bench.build_spec("c2")
    assert spec.n == (101, 101, 101) and spec.nU == 21 ** 3 and spec.idx_np_dtype == np.uint16
    with hjbdp.Backup(spec) as bk:
        inf = bk.info()
        assert inf["kernel_variant"] == 4
        out = bk.solve(100, keep_J=True, keep_idx=True)
        again = bk.solve(100)                                   # the cached graph: same sweep, same bits
    assert out["stages_done"] == 100
    assert np.array_equal(out["J"], again["J"]) and np.array_equal(out["idx"], again["idx"])
    Js, Is = out["J_stages"], out["idx_stages"]
    assert np.array_equal(Js[:, 0], out["J"]) and np.array_equal(Is[:, 0], out["idx"])
    hl, hh = inf["halo_needed_lo"], inf["halo_needed_hi"]
    zero = np.zeros(spec.nS, dtype=np.float32)
    for k_s, p in ((100, 50), (99, 3), (51, 97), (50, 37), (2, 0), (1, 100)):
        prev = zero if k_s == 100 else Js[:, k_s]               # column k_s holds stage k_s + 1
        _plane_from_previous(c_oracle, _abi, spec, prev, Js[:, k_s - 1], Is[:, k_s - 1], (p,), hl, hh)
    G = out["J"].reshape(101, 101, 101, order="F")
    assert G.min() >= 0.0 and G[50, 50, 50] == 0.0
    assert np.all(Js[:, :-1] >= Js[:, 1:] - 1e-5)               # J_k never decreases as the horizon grows


def _c4_deep(env, workload, N, period):
    hjbdp, _abi, c_oracle = env
    import bench
    spec, _ = bench.build_spec(workload)
    assert spec.n == (120,) * 4 and spec.nU == 9
    assert spec.table_dtype == np.float64 and spec.idx_np_dtype == np.uint8           # the bench's exact typing
    assert spec.j_dtype == (np.float16 if workload == "c5" else np.float32)
    mid = N // 2 + 1
    sums = {}

    def oracle_sums(J, idx):
        return c_oracle.monitor_sums(_abi, spec, J, idx, single=True)

    with hjbdp.Backup(spec) as bk:
        inf = bk.info()
        assert inf["kernel_variant"] == 7 and inf["halo_needed_lo"] == 1 and inf["halo_needed_hi"] == 1
        # (1) the reference's sweep: 200 stages, monitor every 50 (Solver_pos_att.m:270-286), single-precision sum
        events = []
        full = bk.solve(N, monitor_period=period, monitor_tol=1e-2, monitor_single=True,
                        progress=lambda k_s, e, e2, sec: events.append((k_s, e, e2)))
        assert full["stages_done"] == N and not full["stopped_early"]
        points = [(N - i * period, 1 + i * period) for i in range(4)]        # (k_s, stages computed when it is reached)
        assert [ev[0] for ev in events] == [k for k, _ in points] and points[2][1] == mid
        # (2) stage pairs: the last two stages and a mid-sweep pair, from sweeps of their own (no monitor: one graph)
        last = full
        for n_done, planes in ((N, (0, 63, 119)), (mid, (17, 104))):
            cur = last if n_done == N else bk.solve(n_done)
            prev = bk.solve(n_done - 1)
            if n_done == N:
                plain = bk.solve(N)                        # the monitored sweep computed the same function
                assert np.array_equal(plain["J"], cur["J"]) and np.array_equal(plain["idx"], cur["idx"])
                del plain
            _plane_from_previous(c_oracle, _abi, spec, prev["J"], cur["J"], cur["idx"], planes, 1, 1)
            if n_done == mid:
                sums[mid] = oracle_sums(cur["J"], cur["idx"])
            del prev, cur
        # (3) the monitor: its sums at the four monitor points are those of J after 1, 1 + period, ... stages
        for n_done in (points[0][1], points[1][1], points[3][1]):
            o = bk.solve(n_done)
            sums[n_done] = oracle_sums(o["J"], o["idx"])
            del o
        f32 = np.float32
        fprev, iprev, expect = 0.0, 0.0, []
        for k_s, n_done in points:
            fs, isum = sums[n_done]
            expect.append((k_s, float(f32(fs) - f32(fprev)), isum - iprev))           # single difference (:276)
            fprev, iprev = fs, isum
        assert events == expect, (events, expect)
        assert full["last_e"] == expect[-1][1] and full["last_e2"] == expect[-1][2]
        # (4) the stop decision at the margin, compared in single (:279): a tolerance one float32 ulp above the third
        # monitor point's |e| stops the sweep at the first monitor point whose |e| is below it; that |e| itself as the
        # tolerance does not stop it there ('<')
        e3 = abs(f32(expect[2][1]))
        for tol in (float(np.nextafter(e3, f32(np.inf))), float(e3)):
            stop = next((k_s for k_s, e, _ in expect if abs(f32(e)) < f32(tol)), None)
            o = bk.solve(N, monitor_period=period, monitor_tol=tol, monitor_single=True)
            assert o["stopped_early"] == (stop is not None), tol
            assert o["stages_done"] == (N if stop is None else N - stop + 1), (tol, stop, o["stages_done"])
            del o
    return full


@pytest.mark.order(3)
@pytest.mark.watchdog(900)
def test_c4_200_stages_deep_bench_typing(env):
    """BASELINE configs[3] exactly as bench.py runs it (float64-built query tables, uint8 labels, axes (x, theta, w, v)),
    200 stages with the reference's monitor."""
    full = _c4_deep(env, "c4", 200, 50)
    assert float(np.min(full["J"])) >= 0.0 and np.all(np.isfinite(full["J"]))


@pytest.mark.order(3)
@pytest.mark.watchdog(900)
def test_c5_100_stages_deep_bench_typing(env):
    """BASELINE configs[4] (float16 cost-to-go storage) in the bench's typing, 100 stages with the monitor every 25.
    Not 200: with binary16 storage this problem's sweep is numerically unstable at the extrapolating corner of the grid -
    rounding noise of relative size 2^-11 between neighbouring cells is amplified by the 7.6-cell extrapolation of w each
    stage: min J turns negative after ~25 stages (-14.5 at 30, -535 at 60, -3.8e4 at 100) and passes binary16's range
    (inf, then NaN) after ~110 (profiles/r04_c5_horizon.log; the float32 sweep stays in [0, 75] over 200 stages).
    Non-finite sweeps are outside the parity contract; up to there the kernel equals the oracle's
    binary16 sweep bit for bit, which is what this test pins."""
    full = _c4_deep(env, "c5", 100, 25)
    assert np.all(np.isfinite(full["J"].astype(np.float32)))


@pytest.mark.order(4)
@pytest.mark.watchdog(900)
def test_c1b_kirk_defaults_199_stages_whole_grid(env):
    """C1b: the committed constructor defaults of Dynamic_Solver (100 x 100 states x 1000 controls, N = 200, single):
    all 199 stages, every state of every stage, against the oracle's sweep."""
    hjbdp, _abi, c_oracle = env
    ds = hjbdp.Dynamic_Solver()
    assert (ds.dx, ds.du, ds.N) == (100, 1000, 200)
    spec = ds.build_spec()
    assert spec.nS == 100 * 100 and spec.nU == 1000 and spec.dtype == np.float32
    with hjbdp.Backup(spec) as bk:
        out = bk.solve(ds.N - 1, keep_J=True, keep_idx=True)
    ref = c_oracle.sweep(_abi, spec, ds.N - 1, keep_J=True, keep_idx=True)
    assert out["stages_done"] == 199
    assert np.array_equal(out["J_stages"], ref["J_stages"])
    assert np.array_equal(out["idx_stages"], ref["idx_stages"])
    ds.run()                                                     # the mirror's run() is that sweep
    lab = ref["idx_stages"].reshape(100, 100, 199, order="F")
    assert np.array_equal(ds.u_star_idx, lab[:, :, 0])


@pytest.mark.order(4)
@pytest.mark.watchdog(900)
def test_attitude_simplified_reference_grid_200_stages_whole_grid(env):
    """Row a10 at the reference's own size: Solver_attitude.simplified_run's 1000 x 300 (w, theta) grid x 3 torques,
    float64, 200 stages, every state of every stage of every channel against the oracle's sweep."""
    hjbdp, _abi, c_oracle = env
    sa = hjbdp.Solver_attitude()
    assert sa.n_mesh_w_simplified == 1000 and sa.n_mesh_t == 300
    sa.simplified_run(n_stages=200)
    assert sa.batch_groups == [1, 1, 1]      # 3 x 3e5 states are more than one round of the wave slots: three chains, not one batch
    for ch in range(3):
        spec, s_w, s_t = sa.build_spec_simplified(ch)
        assert spec.n == (1000, 300) and spec.dtype == np.float64
        ref = c_oracle.sweep(_abi, spec, 200, keep_J=(ch == 0), keep_idx=(ch == 0))
        assert np.array_equal(sa.F_values[ch].reshape(-1, order="F"), ref["J"]), ch
        assert np.array_equal(sa.U_idx[ch].reshape(-1, order="F"), ref["idx"]), ch
        if ch == 0:
            with hjbdp.Backup(spec) as bk:
                out = bk.solve(200, keep_J=True, keep_idx=True)
            assert np.array_equal(out["J_stages"], ref["J_stages"])
            assert np.array_equal(out["idx_stages"], ref["idx_stages"])


# ---------------------------------------------------------------------------------------------------------------------
# The 6-D attitude model (BASELINE configs[2]): attitude-control/Solver_attitude.m:261-300 (the stage loop),
# :384-411 (calculate_J_U_opt_state_M: the 3-level min cascade), :413-506 (the next-state tables).
# ---------------------------------------------------------------------------------------------------------------------
def _attitude_asv(hjbdp, h):
    """Solver_attitude at the size the reference can actually run - the .asv revision's 11^3 (w) x 10^3 (angles) states x
    3^3 torques (Solver_attitude.asv:97,104), its T_final = 1, h = 0.05 -> 19 stages (:116-122,167)."""
    sa = hjbdp.Solver_attitude(n_mesh_w=11, n_mesh_q=10)
    sa.h = h
    return sa


@pytest.mark.order(4)
@pytest.mark.watchdog(900)
@pytest.mark.parametrize("h", [0.05, 0.005])
def test_attitude_run_reference_size_19_stages_whole_grid(env, h):
    """Row a11 at the only size the reference can run: Solver_attitude.run on 11^3 x 10^3 x 27 over its 19 stages, EVERY
    state of EVERY stage (J and labels, bit for bit) against the oracle's sweep, in the three forms the library serves it:
    the reference's axis order (w1, w2, w3, yaw, pitch, roll - the general table kernel), relabelled with the angles first
    (K3's window modes on tabulated next angles) and with the on-the-fly quaternion model (the kernel mode C3 runs).
    h = 0.05 is the .asv's step: a torque step moves w by 1.1 - 1.3 cells, so sweeps cross several cells and leave the
    prepared window (four-plane window modes 2 / 3, synchronous far-cell gathers); h = 0.005 is the committed .m's step
    (three-plane window modes 5 / 6)."""
    hjbdp, _abi, c_oracle = env
    sa = _attitude_asv(hjbdp, h)
    full = sa.build_spec_full()
    relab, to_old = hjbdp.permute_state_axes(full, sa.AXIS_ORDER)
    model = sa.build_spec_model()
    assert full.n == (11, 11, 11, 10, 10, 10) and relab.n == model.n == (10, 10, 10, 11, 11, 11) and full.nU == 27
    modes = {}
    finals = {}
    for name, spec in (("reference_order", full), ("relabelled", relab), ("on_the_fly", model)):
        if name == "reference_order" and h != 0.05:
            continue                                                 # the table kernel's paths do not depend on the step: once
        with hjbdp.Backup(spec) as bk:
            inf = bk.info()
            modes[name] = (inf["kernel_variant"], bk.get_option("packed2_mode"))
            out = bk.solve(19, keep_J=True, keep_idx=True)
        ref = c_oracle.sweep(_abi, spec, 19, keep_J=True, keep_idx=True)
        assert out["stages_done"] == 19
        bad = np.flatnonzero((out["J_stages"] != ref["J_stages"]).any(axis=0))
        assert bad.size == 0, (name, "J differs in stage columns", bad)
        bad = np.flatnonzero((out["idx_stages"] != ref["idx_stages"]).any(axis=0))
        assert bad.size == 0, (name, "labels differ in stage columns", bad)
        assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
        assert np.all(np.isfinite(out["J"])) and out["J"].min() >= 0
        assert len(np.unique(out["idx_stages"])) >= 20, name        # (nearly) every torque triple is optimal somewhere: a real argmin
        finals[name] = out
    assert h != 0.05 or modes["reference_order"][0] != 4             # not the relabelled kernel
    assert modes["relabelled"] == (4, 2 if h == 0.05 else 5), modes
    assert modes["on_the_fly"] == (4, 3 if h == 0.05 else 6), modes
    # the three forms compute one function: equal to rounding (lerp order / polynomial atan2 differ), same policy almost everywhere
    base = "reference_order" if h == 0.05 else "relabelled"
    a, ia = finals[base]["J"], finals[base]["idx"]
    if base == "relabelled":
        a, ia = to_old(a), to_old(ia)
    for name in ("relabelled", "on_the_fly"):
        b = to_old(finals[name]["J"])
        assert np.max(np.abs(a - b)) <= 1e-4 * np.max(np.abs(a)), name
        assert np.mean(ia == to_old(finals[name]["idx"])) > 0.99, name
    # the mirror's run() is that sweep (U_i_Opt = U_vector(idx), :296-298)
    sa.run(n_stages=19, relabel=True)
    assert np.array_equal(sa.F_values.reshape(-1, order="F"), to_old(finals["relabelled"]["J"]))
    lab = (sa.U_idx[0] - 1) + 3 * (sa.U_idx[1] - 1) + 9 * (sa.U_idx[2] - 1) + 1
    assert np.array_equal(lab.reshape(-1, order="F"), to_old(finals["relabelled"]["idx"]))


def _sample_6d(spec, rng, n_random, cells_of_last=None):
    """A sample of linear state indices: random states, every grid corner, edge mid-points, both sides of 256-state chunk
    boundaries (K3's unit of work) and of plane boundaries, the first and the last state."""
    n = np.array(spec.n, dtype=np.int64)
    stride = np.concatenate([[1], np.cumprod(n[:-1])])
    lin = lambda cs: int(np.dot(np.array(cs, dtype=np.int64), stride))
    pts = [lin([(c >> a) & 1 and n[a] - 1 for a in range(6)]) for c in range(64)]                 # the 64 corners
    pts += [lin([n[a] // 2 if a != b else e for a in range(6)]) for b in range(6) for e in (0, n[b] - 1)]   # face centres
    chunks = rng.integers(1, spec.nS // 256, 40)
    pts += [int(c) * 256 + d for c in chunks for d in (-1, 0)]
    planes = rng.integers(1, n[5], 6)
    pts += [int(p) * int(stride[5]) + d for p in planes for d in (-1, 0)]
    pts += [0, spec.nS - 1]
    return np.unique(np.concatenate([rng.integers(0, spec.nS, n_random), np.array(pts, dtype=np.int64)]))


def _states_leaving_the_window(spec, pool):
    """Of the pool, the states whose innermost-control sweep of the last axis (w3 + h(c w1 w2 + U3 / J3), :425) visits
    three or more cells for at least one (U1, U2): their steps leave K3's two-cell window (the synchronous gather path)."""
    n = np.array(spec.n, dtype=np.int64)
    idx = np.stack(np.unravel_index(pool, spec.n, order="F"), axis=0)
    k = np.asarray(spec.knots[5], dtype=np.float32)
    base, t3 = spec.next_terms[5][0].data, spec.next_terms[5][1].data            # (w3), (w1, w2, U3)
    q = (np.asarray(base, dtype=np.float32)[idx[5]][:, None] + np.asarray(t3, dtype=np.float32)[idx[3], idx[4], :]).astype(np.float32)
    cell = np.clip(np.searchsorted(k, q, side="right") - 1, 0, n[5] - 2)
    return pool[(cell.max(axis=1) - cell.min(axis=1)) >= 2]


@pytest.mark.order(5)
@pytest.mark.watchdog(900)
@pytest.mark.parametrize("form,h", [("on_the_fly", 0.005), ("tabulated", 0.005), ("on_the_fly", 0.03)])
def test_6d_24_pow_6_five_stages_deep(env, form, h):
    """The 6-D model on 24^6 = 1.9e8 states x 11^3 torques, FIVE stages deep from a zero terminal cost (the reference's
    start, :271-279), device-resident ping-pong.  After every stage the oracle recomputes >= 10^4 sampled states of stage k
    from the GPU's OWN stage k+1 (downloaded, 764 MB) - corners, faces, both sides of chunk and plane boundaries, random
    states and, for the coarse-step case, hundreds of states whose torque sweep leaves the two-cell window - bit for bit.
    h = 0.03: a torque step moves w3 by 0.36 of a cell (three-plane window still admitted), the sweep of 11 spans 3.6 cells."""
    hjbdp, _abi, c_oracle = env
    sa = hjbdp.Solver_attitude(n_mesh_w=24, n_mesh_q=24)
    sa.U_vector = np.linspace(-0.11, 0.11, 11)
    sa.h = h
    if form == "tabulated":
        spec0, _ = hjbdp.permute_state_axes(sa.build_spec_full(), sa.AXIS_ORDER)
    else:
        spec0 = sa.build_spec_model()
    spec = hjbdp.ProblemSpec(spec0.knots, spec0.m, spec0.next_terms, spec0.cost_terms, dtype=np.float32, index_base=spec0.index_base,
                             model=spec0.model, idx_dtype="auto")
    assert spec.nS == 24 ** 6 and spec.nU == 1331 and spec.idx_np_dtype == np.uint16
    rng = np.random.default_rng(2406)
    sel = _sample_6d(spec, rng, 10000)
    far = _states_leaving_the_window(spec, rng.integers(0, spec.nS, 200000))
    if h >= 0.03:
        assert far.size > 1000
        sel = np.unique(np.concatenate([sel, far[:2000]]))
    else:
        assert far.size == 0                                    # the committed step: every sweep stays within two cells
    assert sel.size >= 10000
    dA, dB = hjbdp.DeviceBuffer(spec.nS * 4), hjbdp.DeviceBuffer(spec.nS * 4)
    dI = hjbdp.DeviceBuffer(spec.nS * 2)
    try:
        with hjbdp.Backup(spec) as bk:
            assert bk.info()["kernel_variant"] == 4
            # h = 0.005: K15 (kernels_uniwin.h, modes 7 / 8); h = 0.03: its plan flags the sweeps that span 3.6 cells and K3's mode 6 stays
            assert bk.get_option("packed2_mode") == ((8 if form == "on_the_fly" else 7) if h < 0.03 else 6)
            J_prev = np.zeros(spec.nS, dtype=np.float32)
            dA.upload(J_prev)
            src, dst = dA, dB
            labels_seen = set()
            for stage in range(5):
                bk.backup_stage_device(src, dst, dI)
                bk.check_device_status()
                Jr, ir = c_oracle.backup_states_from_J(_abi, spec, J_prev, sel)
                Jg, ig = dst.gather(np.float32, sel), dI.gather(np.uint16, sel)
                bad = np.flatnonzero(Jg != Jr)
                assert bad.size == 0, (stage, "J", sel[bad[:5]], Jg[bad[:5]], Jr[bad[:5]])
                bad = np.flatnonzero(ig != ir)
                assert bad.size == 0, (stage, "labels", sel[bad[:5]], ig[bad[:5]], ir[bad[:5]])
                labels_seen.update(np.unique(ig).tolist())
                J_prev = dst.download(np.float32)
                assert np.array_equal(J_prev[sel], Jg)
                src, dst = dst, src
            assert np.all(np.isfinite(J_prev)) and J_prev.min() >= 0 and J_prev.max() > 0
            assert len(labels_seen) > 100                       # stage 5 of a real sweep: many distinct torque triples win
    finally:
        dA.free(); dB.free(); dI.free()


@pytest.mark.order(6)
@pytest.mark.watchdog(900)
def test_c3_full_size_second_stage_whole_grid_and_as_eight_slabs(env):
    """BASELINE C3 (51^6 x 11^3, 176 GB resident) TWO stages deep, twice: as ONE launch over the whole grid, and as the EIGHT w3
    slabs of 6 - 7 planes `bench.py --gpus 8 --workload c3` cuts it into (hjb_rank_create / hjb_rank_stage: interior + boundary strips
    per slab, the halo plane of 1.38 GB per neighbour moved between the slabs' buffers by device-to-device copies - the part RCCL plays
    on eight GPUs; all eight on this box's one GPU, 215 GB resident).  Stage 1 from a separable cost-to-go built on the device, stage 2
    from the GPU's own stage-1 output - a min over 1331 torque triples, not separable.  The oracle lists every element of stage 1 the
    sampled states' backups read (all controls x 64 corners), those elements are gathered from the device (hjb_device_gather) and the
    oracle performs the backup on that sample; J and labels of stage 2 must agree bit for bit, in both forms.  The sample holds both
    sides of every slab boundary, of 256-state chunk boundaries, and the indices around 2^31, 2^32, 2^33 and 2^34 (64-bit indexing +
    slab offsets).  Stage 1 itself is pinned on the same sample by the separable checker.  (attitude-control/Solver_attitude.m:280-287)"""
    hjbdp, _abi, c_oracle = env
    free, total = hjbdp.device_mem_info(0)
    if free < 225 * 2 ** 30:
        pytest.skip("needs 225 GB of free HBM, have %.0f GB" % (free / 2 ** 30))
    sa = hjbdp.Solver_attitude(n_mesh_w=51, n_mesh_q=51)
    sa.U_vector = np.linspace(-0.11, 0.11, 11)
    spec0 = sa.build_spec_model()
    spec = hjbdp.ProblemSpec(spec0.knots, spec0.m, spec0.next_terms, spec0.cost_terms, dtype=np.float32, index_base=spec0.index_base,
                             model=spec0.model, idx_dtype="auto")
    assert spec.nS == 51 ** 6 and spec.nU == 1331 and spec.idx_np_dtype == np.uint16
    rng = np.random.default_rng(5102)
    vecs = [(rng.random(n) * (1.0 + a)).astype(np.float32) for a, n in enumerate(spec.n)]
    inner = spec.nS // 51                                       # states per w3 plane: 51^5
    world = 8
    ranks = [hjbdp.RankSlab(spec, 0, r, world, overlap=True) for r in range(world)]
    try:
        assert ranks[0].begin == 0 and ranks[-1].end == 51 and all(a.end == b.begin for a, b in zip(ranks, ranks[1:]))
        assert sorted(r.end - r.begin for r in ranks) == [6, 6, 6, 6, 6, 7, 7, 7] and all(r.split for r in ranks)
        assert all(r.halo_lo == (1 if i else 0) and r.halo_hi == (1 if i < world - 1 else 0) for i, r in enumerate(ranks))
        sel = _sample_6d(spec, rng, 300)
        edge = [r.begin * inner + d for r in ranks[1:] for d in (-1, 0, 255, 256, -inner, inner - 1)]      # both sides of every slab boundary
        sel = np.unique(np.concatenate([sel, edge, [2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32, 2 ** 33 + 255, 2 ** 33 + 256, 2 ** 34 - 1]]))
        keys = c_oracle.backup_states_touch(_abi, spec, sel)
        assert keys.size > 64 * sel.size // 8 and keys.max() < spec.nS
        J1r, i1r = c_oracle.backup_states(_abi, spec, vecs, sel)
        # ---- the whole grid as one launch ----------------------------------------------------------------------------------------
        dA, dB = hjbdp.DeviceBuffer(spec.nS * 4), hjbdp.DeviceBuffer(spec.nS * 4)
        dI = hjbdp.DeviceBuffer(spec.nS * 2)
        try:
            with hjbdp.Backup(spec) as bk:
                assert bk.info()["kernel_variant"] == 4 and bk.get_option("packed2_mode") == 8        # K15 (kernels_uniwin.h)
                bk.fill_separable(vecs, dA)
                bk.backup_stage_device(dA, dB, dI)                  # stage 1: separable -> dB
                bk.check_device_status()
                assert np.array_equal(dB.gather(np.float32, sel), J1r) and np.array_equal(dI.gather(np.uint16, sel), i1r)
                vals = dB.gather(np.float32, keys)
                bk.backup_stage_device(dB, dA, dI)                  # stage 2: the GPU's own stage 1 -> dA (over the separable fill)
                bk.check_device_status()
            J2g, i2g = dA.gather(np.float32, sel), dI.gather(np.uint16, sel)
        finally:
            dA.free(); dB.free(); dI.free()
        J2r, i2r = c_oracle.backup_states_sparse(_abi, spec, keys, vals, sel)
        bad = np.flatnonzero(J2g != J2r)
        assert bad.size == 0, ("J", sel[bad[:5]], J2g[bad[:5]], J2r[bad[:5]])
        bad = np.flatnonzero(i2g != i2r)
        assert bad.size == 0, ("labels", sel[bad[:5]], i2g[bad[:5]], i2r[bad[:5]])
        assert not np.array_equal(J2g, J1r)                    # a second stage happened
        # ---- the same two stages as eight slabs --------------------------------------------------------------------------------
        lib = ranks[0].lib
        pb = inner * 4
        bufs = []
        try:
            for r in ranks:
                planes = r.end - r.begin + r.halo_lo + r.halo_hi
                bufs.append(([hjbdp.DeviceBuffer(pb * planes), hjbdp.DeviceBuffer(pb * planes)], hjbdp.DeviceBuffer(inner * (r.end - r.begin) * 2)))

            def slab_of(keys_global):
                """global state indices -> (slab number, element offset in the slab's haloed J buffer, offset in its label buffer)"""
                pl = keys_global // inner
                which = np.searchsorted([r.end for r in ranks], pl, side="right")
                b0 = np.array([r.begin for r in ranks])[which]
                hl = np.array([r.halo_lo for r in ranks])[which]
                return which, keys_global - b0 * inner + hl * inner, keys_global - b0 * inner

            def gather(which_buf, keys_global, dtype, labels=False):
                w, oj, oi = slab_of(np.asarray(keys_global, dtype=np.int64))
                out = np.empty(len(w), dtype=dtype)
                for i in range(world):
                    m = w == i
                    if m.any():
                        out[m] = (bufs[i][1] if labels else bufs[i][0][which_buf]).gather(dtype, (oi if labels else oj)[m])
                return out

            def exchange(cur):
                for i, r in enumerate(ranks):                     # my halo planes from my neighbours' owned boundary planes
                    if r.halo_lo:
                        lo = ranks[i - 1]
                        assert lib.hjb_device_copy(0, bufs[i][0][cur].ptr, bufs[i - 1][0][cur].ptr + pb * (lo.halo_lo + lo.end - lo.begin - 1), pb, _abi.HJB_COPY_D2D) == 0
                    if r.halo_hi:
                        hi = ranks[i + 1]
                        assert lib.hjb_device_copy(0, bufs[i][0][cur].ptr + pb * (r.halo_lo + r.end - r.begin), bufs[i + 1][0][cur].ptr + pb * hi.halo_lo, pb, _abi.HJB_COPY_D2D) == 0
            for i, r in enumerate(ranks):
                r.fill_separable(vecs, bufs[i][0][0])             # owned planes AND halos of the terminal cost
            for i, r in enumerate(ranks):
                r.stage(bufs[i][0][0], bufs[i][0][1], bufs[i][1])                                 # stage 1
            for r in ranks:
                r.check_device_status()
            assert np.array_equal(gather(1, sel, np.float32), J1r) and np.array_equal(gather(1, sel, np.uint16, labels=True), i1r)
            assert np.array_equal(gather(1, keys, np.float32), vals)      # every element the second stage will read = the whole-grid launch's
            exchange(1)
            for i, r in enumerate(ranks):
                r.stage(bufs[i][0][1], bufs[i][0][0], bufs[i][1])                                 # stage 2 from the slabs' own stage 1
            for r in ranks:
                r.check_device_status()
            J2s, i2s = gather(0, sel, np.float32), gather(0, sel, np.uint16, labels=True)
            bad = np.flatnonzero(J2s != J2r)
            assert bad.size == 0, ("slabs J", sel[bad[:5]], J2s[bad[:5]], J2r[bad[:5]])
            bad = np.flatnonzero(i2s != i2r)
            assert bad.size == 0, ("slabs labels", sel[bad[:5]], i2s[bad[:5]], i2r[bad[:5]])
        finally:
            for J, I in bufs:
                J[0].free(); J[1].free(); I.free()
    finally:
        for r in ranks:
            r.close()
