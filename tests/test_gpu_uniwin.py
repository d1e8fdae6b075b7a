"""GPU parity tests (-m gpu) of K15 (csrc/kernels_uniwin.h, variant 4 modes 7 / 8): the window kernel for chunks that share
their rate axes - Solver_attitude.run's shape (attitude-control/Solver_attitude.m:261-300, 400-409, 413-506) with the angle
axes first.  Whole grids against the oracle's sweep, bit for bit, and against K3's window modes 5 / 6 on the same handle."""
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


CASES = [
    # n_so, n_rates, m, gain, nonuniform
    ((8, 6, 6), (4, 4, 5), (11, 11, 11), (0.55, 0.5, 0.6), False),      # the attitude shape: every axis crosses one cell boundary
    ((8, 6, 6), (5, 4, 4), (3, 5, 12), (0.2, 0.25, 0.25), True),         # twelve inner controls, non-uniform knots, few outer levels
    ((20, 13), (4, 5, 4), (11, 4, 11), (0.6, 0.2, 0.5), False),         # D = 5
    ((300,), (5, 4, 6), (6, 11, 12), (0.25, 0.2, 0.02), True),         # D = 4; the last axis never changes cell
]


@pytest.mark.parametrize("n_so,n_rates,m,gain,nonuniform", CASES)
def test_uniwin_whole_grid_bit_exact(env, n_so, n_rates, m, gain, nonuniform):
    hjbdp, _abi, c_oracle = env
    from problems import rate_shared_problem, random_terminal
    spec = rate_shared_problem(1500 + len(n_so), n_so, n_rates, m=m, gain=gain, nonuniform=nonuniform)
    term = random_terminal(spec, 21)
    ref = c_oracle.sweep(_abi, spec, 2, terminal=term, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["kernel_variant"] == 4
        assert bk.get_option("uniwin_ok") == 1 and bk.get_option("uniwin_slow_points") == 0
        assert bk.get_option("packed2_mode") == 7 and bk.get_option("uniwin") == 1
        out = bk.solve(2, terminal=term, keep_J=True, keep_idx=True)
        bk.set_option("uniwin", 0)                              # K3's window mode on the same handle: the same bits
        assert bk.get_option("packed2_mode") == 5
        old = bk.solve(2, terminal=term, keep_J=True, keep_idx=True)
        bk.set_option("uw_tile", 1 + 8 * 1 + 64 * 0)            # another tiling of the chunk walk: the same bits
        bk.set_option("uniwin", 1)
        tiled = bk.solve(2, terminal=term, keep_J=True, keep_idx=True)
        bk.set_option("uw_block", 64)                           # one wave per workgroup, 64-state chunks: the same bits
        assert bk.get_option("uw_block") == 64
        small = bk.solve(2, terminal=term, keep_J=True, keep_idx=True)
    for name, o in (("uniwin", out), ("mode 5", old), ("tile 2x2x1", tiled), ("64-state chunks", small)):
        bad = np.flatnonzero(o["J_stages"] != ref["J_stages"])
        assert bad.size == 0, (name, "J", bad[:8])
        bad = np.flatnonzero(o["idx_stages"] != ref["idx_stages"])
        assert bad.size == 0, (name, "labels", bad[:8])
    assert len(np.unique(out["idx_stages"])) > 8


@pytest.mark.parametrize("n,m,order", [((130, 5, 4, 5), (3, 4, 11), "std"), ((130, 4, 5, 4), (2, 5, 12), "l1_first"),
                                       ((20, 7, 3, 4, 5), (3, 6, 11), "l0_first"), ((20, 7, 4, 3, 5), (4, 3, 12), "inner_first")])
def test_uniwin_cost_term_orders(env, n, m, order):
    """A control's cost term FIRST in the canonical sum (no state terms before it): the level-0 / level-1 / inner term leads, on
    the chain problem of the K3 trip-shape tests (tests/test_gpu_parity.py) with eleven / twelve inner controls."""
    hjbdp, _abi, c_oracle = env
    from problems import random_terminal
    from test_gpu_parity import _chain_spec
    D = len(n)
    orders = {"std": lambda k: list(range(k)), "l0_first": lambda k: [D, D + 1, D + 2], "l1_first": lambda k: [D + 1, D + 2],
              "inner_first": lambda k: [0, D + 2]}
    spec = _chain_spec(n, m, (0.05, 0.05, 0.10), -1.0, 1.0, orders[order])
    term = random_terminal(spec, 5)
    ref = c_oracle.sweep(_abi, spec, 2, terminal=term, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec, variant=4) as bk:
        assert bk.get_option("packed2_mode") == 7, (bk.get_option("uniwin_ok"), bk.get_option("uniwin_slow_points"))
        out = bk.solve(2, terminal=term, keep_J=True, keep_idx=True)
    assert np.array_equal(out["J_stages"], ref["J_stages"]) and np.array_equal(out["idx_stages"], ref["idx_stages"])


def test_uniwin_long_sweep_through_graph_replay_and_both_walks(env):
    """50 stages through hjb_solve (the ping-pong loop replayed from a hipGraph: the per-launch reset of the walk's claim counters is a
    memset node of that graph), with the chunk walk claimed from per-XCD counters (default) and with the fixed stride: every stage
    equal to the oracle's."""
    hjbdp, _abi, c_oracle = env
    from problems import rate_shared_problem
    spec = rate_shared_problem(31, (140,), (4, 3, 4), m=(3, 4, 11), gain=(0.3, 0.2, 0.25))
    ref = c_oracle.sweep(_abi, spec, 50, keep_J=True, keep_idx=True)          # zero terminal cost (the reference's start)
    with hjbdp.Backup(spec) as bk:
        assert bk.get_option("packed2_mode") == 7 and bk.get_option("uw_claim") == 1
        for claim in (1, 0, 1):
            bk.set_option("uw_claim", claim)
            out = bk.solve(50, keep_J=True, keep_idx=True)
            assert np.array_equal(out["J_stages"], ref["J_stages"]) and np.array_equal(out["idx_stages"], ref["idx_stages"]), claim
        bk.set_option("graph", 0)
        out = bk.solve(50)
        assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
    assert np.all(np.isfinite(ref["J_stages"])) and len(np.unique(ref["idx"])) > 5


def test_uniwin_points_outside_the_usual_shape_take_the_slow_path(env):
    """A rate step of 0.3 cells per control level (the three-plane window is still admitted) carries a sweep of eleven across
    three cells: nearly every point is flagged, the automatic choice stays with mode 5, and K15 forced on must still agree with
    the oracle bit for bit through its per-backup path."""
    hjbdp, _abi, c_oracle = env
    from problems import rate_shared_problem, random_terminal
    spec = rate_shared_problem(77, (8, 6, 6), (4, 3, 4), m=(11, 11, 11), gain=(1.4, 0.5, 1.6))
    term = random_terminal(spec, 4)
    ref = c_oracle.sweep(_abi, spec, 2, terminal=term, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec) as bk:
        assert bk.get_option("uniwin_ok") == 1 and 10 < bk.get_option("uniwin_slow_points") <= 48
        assert bk.get_option("packed2_mode") == 5 and bk.get_option("uniwin") == 0
        bk.set_option("uniwin", 1)
        assert bk.get_option("packed2_mode") == 7
        out = bk.solve(2, terminal=term, keep_J=True, keep_idx=True)
    assert np.array_equal(out["J_stages"], ref["J_stages"]) and np.array_equal(out["idx_stages"], ref["idx_stages"])


@pytest.mark.parametrize("form", ["tabulated", "on_the_fly"])
def test_uniwin_attitude_model_slab_and_float16(env, form):
    """The attitude model itself (tabulated next angles: mode 7; HJB_MODEL_QUAT_EULER321: mode 8) on 8^3 angles x 6^3 rates x
    11^3 torques: whole grid, a slab of the last axis with halos, and float16 cost-to-go storage - each against the oracle."""
    hjbdp, _abi, c_oracle = env
    from problems import random_terminal
    sa = hjbdp.Solver_attitude(n_mesh_w=6, n_mesh_q=8)
    sa.U_vector = np.linspace(-0.11, 0.11, 11)
    if form == "tabulated":
        spec, _ = hjbdp.permute_state_axes(sa.build_spec_full(), sa.AXIS_ORDER)
    else:
        spec = sa.build_spec_model()
    mode = 7 if form == "tabulated" else 8
    term = random_terminal(spec, 8)
    ref = c_oracle.sweep(_abi, spec, 2, terminal=term, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec) as bk:
        assert bk.get_option("packed2_mode") == mode
        out = bk.solve(2, terminal=term, keep_J=True, keep_idx=True)
    assert np.array_equal(out["J_stages"], ref["J_stages"]) and np.array_equal(out["idx_stages"], ref["idx_stages"])
    # a slab: planes 2 .. 4 of w3 with one halo plane each way
    nl = spec.n[-1]
    inner = spec.nS // nl
    Jw, iw = c_oracle.backup_stage(_abi, spec, term)
    T2 = term.reshape(inner, nl, order="F")
    b, e, lo, hi = 2, 5, 1, 1
    with hjbdp.Backup(spec, slab=(b, e, lo, hi)) as bk:
        assert bk.get_option("packed2_mode") == mode
        Jo, io = bk.backup_stage(np.asfortranarray(T2[:, b - lo:e + hi]).reshape(-1, order="F"))
    assert np.array_equal(Jo.reshape(inner, -1, order="F")[:, lo:lo + e - b], Jw.reshape(inner, nl, order="F")[:, b:e])
    assert np.array_equal(io, iw.reshape(inner, nl, order="F")[:, b:e].reshape(-1, order="F"))
    # float16 storage of the cost-to-go
    s16 = hjbdp.ProblemSpec(spec.knots, spec.m, spec.next_terms, spec.cost_terms, dtype=np.float32, index_base=spec.index_base,
                            model=spec.model, j_storage="float16")
    t16 = term.astype(np.float16)
    r16 = c_oracle.sweep(_abi, s16, 2, terminal=t16, keep_J=True, keep_idx=True)
    with hjbdp.Backup(s16) as bk:
        assert bk.get_option("packed2_mode") == mode
        o16 = bk.solve(2, terminal=t16, keep_J=True, keep_idx=True)
    assert np.array_equal(o16["J_stages"], r16["J_stages"]) and np.array_equal(o16["idx_stages"], r16["idx_stages"])


@pytest.mark.order(6)
@pytest.mark.watchdog(300)
def test_uniwin_randomised_stress_slice(env):
    """Ten seconds of tools/stress_uniwin.py: random rate-shared shapes (D = 4 .. 6, gains from sub-cell to several cells - the plain
    path of points outside the two-cell windows included -, non-uniform knots, float16 storage, slabs, the walk options); every value
    and label equals the oracle's.  (The long form: profiles/r06_stress_uniwin.txt.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_uniwin.py"), "10", "7"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "stress ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
