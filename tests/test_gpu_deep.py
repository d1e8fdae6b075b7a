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
def test_c5_200_stages_deep_bench_typing(env):
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
