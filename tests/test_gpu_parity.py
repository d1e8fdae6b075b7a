"""GPU parity tests (-m gpu): the HIP path through the C ABI against the oracle.

Bars: J and argmin BIT-EXACT against the C twin (oracle/hjb_oracle.c, same
canonical arithmetic); <= 1e-6 relative (north_star's tolerance; observed ~1e-13)
against the MATLAB fixture test/obj_1.mat for the float64 Kirk problem."""
from pathlib import Path

import numpy as np
import pytest

ROOT_GOLDEN = Path(__file__).resolve().parent / "golden"

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import hjbdp
    from hjbdp import _abi
    from oracle import c_oracle
    if hjbdp.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run the HIP path (no fallback)")
    return hjbdp, _abi, c_oracle


def _kirk(hjbdp, precision, N, dx, du):
    ds = hjbdp.Dynamic_Solver(precision=precision)
    ds.N, ds.dx, ds.du = N, dx, du
    return ds


@pytest.mark.order(1)
def test_kirk_fixture_full_sweep_vs_matlab(env, golden):
    """C1a: exactly test/obj_1.txt (35x35x100, N=130, f64): all 129 stages."""
    hjbdp, _abi, c_oracle = env
    ds = _kirk(hjbdp, "double", 130, 35, 100)
    ds.run()
    ref = golden["J_star"]
    assert ds.J_star.shape == ref.shape
    rel = np.max(np.abs(ds.J_star[:, :, :129] - ref[:, :, :129]) / np.abs(ref[:, :, :129]))
    assert rel <= 1e-6, rel          # north_star tolerance
    assert rel <= 1e-12, rel         # what the restatement actually achieves
    assert not ds.J_star[:, :, 129].any() and not ds.u_star[:, :, 129].any()
    u_ref = golden["U_mesh"][golden["u_star_idx"]]
    assert np.array_equal(ds.u_star[:, :, :129], u_ref)   # no argmin flips (gap >= 1e-5)
    X, U = ds.get_optimal_path()
    assert np.max(np.abs(X - golden["traj_X"])) < 1e-9
    assert np.max(np.abs(U - golden["traj_U"])) < 1e-9


@pytest.mark.order(1)
def test_kirk_fixture_bit_exact_vs_oracle(env):
    hjbdp, _abi, c_oracle = env
    spec = _kirk(hjbdp, "double", 130, 35, 100).build_spec()
    with hjbdp.Backup(spec) as bk:
        out = bk.solve(129, keep_J=True, keep_idx=True)
    ref = c_oracle.sweep(_abi, spec, 129, keep_J=True, keep_idx=True)
    assert np.array_equal(out["J_stages"], ref["J_stages"])
    assert np.array_equal(out["idx_stages"], ref["idx_stages"])
    assert out["stages_done"] == 129 and not out["stopped_early"]


@pytest.mark.order(1)
def test_kirk_single_bit_exact_vs_oracle(env):
    """C1b typing (single tables) at a size the oracle finishes in seconds."""
    hjbdp, _abi, c_oracle = env
    spec = _kirk(hjbdp, "single", 30, 60, 250).build_spec()
    with hjbdp.Backup(spec) as bk:
        out = bk.solve(29, keep_J=True, keep_idx=True)
    ref = c_oracle.sweep(_abi, spec, 29, keep_J=True, keep_idx=True)
    assert np.array_equal(out["J_stages"], ref["J_stages"])
    assert np.array_equal(out["idx_stages"], ref["idx_stages"])


SHAPES = [
    # (n, m, dtype, nonuniform)
    ((9,), (5,), np.float64, False),
    ((2, 2), (1,), np.float64, False),                 # minimum sizes
    ((17, 13), (7,), np.float64, True),
    ((17, 13), (4, 3), np.float32, False),
    ((11, 9, 8), (5, 4, 3), np.float32, True),
    ((11, 9, 8), (6,), np.float64, False),
    ((7, 6, 5, 4), (9,), np.float32, True),
    ((6, 5, 4, 5), (3, 2), np.float64, False),
    ((5, 4, 3, 4, 3), (3, 3), np.float32, False),
    ((4, 3, 4, 3, 3, 4), (3, 3, 3), np.float32, False),  # attitude shape (6-D x 3-D)
    ((4, 3, 4, 3, 3, 4), (2, 2, 2), np.float64, True),
]


@pytest.mark.parametrize("variant", [None, 0, 3, 5])
@pytest.mark.parametrize("n,m,dtype,nonuniform", SHAPES)
def test_random_problems_bit_exact(env, n, m, dtype, nonuniform, variant):
    hjbdp, _abi, c_oracle = env
    from problems import random_problem, random_terminal
    spec = random_problem(1234 + len(n) * 10 + len(m), n, m, dtype=dtype, nonuniform=nonuniform, index_base=1)
    term = random_terminal(spec, 7)
    with hjbdp.Backup(spec, variant=variant) as bk:
        out = bk.solve(3, terminal=term, keep_J=True, keep_idx=True)
        J1, i1 = bk.backup_stage(term)
    ref = c_oracle.sweep(_abi, spec, 3, terminal=term, keep_J=True, keep_idx=True)
    assert np.array_equal(out["J_stages"], ref["J_stages"])
    assert np.array_equal(out["idx_stages"], ref["idx_stages"])
    assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
    # single-stage entry point = first computed stage (k_s = 3 -> column 2)
    assert np.array_equal(J1, ref["J_stages"][:, 2]) and np.array_equal(i1, ref["idx_stages"][:, 2])
    assert out["idx"].min() >= 1 and out["idx"].max() <= spec.nU


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,dtype,nonuniform", SHAPES)
def test_table_kernel_both_index_forms_bit_exact(env, n, m, dtype, nonuniform):
    """Variant 5 runs in 32-bit index arithmetic when every index fits 31 bits (k_backup_tabled32: hoisted corner offsets, axis-0 corner
    pairs as one load, the next control's table entries one control ahead) and in the general 64-bit form otherwise.  Both forms, forced
    by the option, must equal the oracle bit for bit - on whole grids and on a slab - and the small problems of this suite must get the
    32-bit form by default."""
    hjbdp, _abi, c_oracle = env
    from problems import random_problem, random_terminal
    spec = random_problem(4321 + len(n) * 10 + len(m), n, m, dtype=dtype, nonuniform=nonuniform, index_base=1)
    term = random_terminal(spec, 9)
    ref = c_oracle.sweep(_abi, spec, 3, terminal=term, keep_J=True, keep_idx=True)
    for form in (1, 0):
        with hjbdp.Backup(spec, variant=5) as bk:
            assert bk.get_option("tabled_i32") == 1                    # the default on a small problem
            bk.set_option("tabled_i32", form)
            assert bk.get_option("tabled_i32") == form
            out = bk.solve(3, terminal=term, keep_J=True, keep_idx=True)
        assert np.array_equal(out["J_stages"], ref["J_stages"]) and np.array_equal(out["idx_stages"], ref["idx_stages"]), form
    nl = n[-1]
    if nl >= 6:
        b, e = nl // 3, nl - nl // 3
        with hjbdp.Backup(spec, variant=5) as bk:
            need = bk.info()
        hl, hh = min(need["halo_needed_lo"], b), min(need["halo_needed_hi"], nl - e)
        inner = spec.nS // nl
        sub = np.asfortranarray(term.reshape(inner, -1, order="F")[:, b - hl:e + hh]).reshape(-1, order="F")
        Jr, ir = c_oracle.backup_stage(_abi, spec, sub, slab=(b, e, hl, hh))
        for form in (1, 0):
            with hjbdp.Backup(spec, slab=(b, e, hl, hh), variant=5) as bk:
                bk.set_option("tabled_i32", form)
                Jg, ig = bk.backup_stage(sub)
            assert np.array_equal(Jg, Jr) and np.array_equal(ig, ir), form


NESTED = [
    ((9, 8), (3,), np.float64, False, False),              # Solver_position / attitude-simplified shape
    ((13, 11), (7,), np.float32, True, True),
    ((9, 8, 7), (5, 4, 3), np.float32, False, False),      # C2 shape
    ((9, 8, 7), (4, 5), np.float64, True, True),
    ((6, 5, 4, 5), (3, 4), np.float32, False, False),
    ((4, 3, 4, 3, 3, 5), (3, 3, 3), np.float32, False, False),  # C3 shape
    ((4, 3, 4, 3, 3, 5), (2, 3, 2), np.float64, True, True),
]


@pytest.mark.parametrize("n,m,dtype,nonuniform,mixed", NESTED)
def test_nested_variant_bit_exact(env, n, m, dtype, nonuniform, mixed):
    """Variant 1 (control-nested) must be selected for spacecraft-shaped problems
    and agree bit for bit with the oracle and with the generic kernel."""
    hjbdp, _abi, c_oracle = env
    from problems import nested_problem, random_terminal
    spec = nested_problem(4321 + len(n), n, m, dtype=dtype, nonuniform=nonuniform, mixed_inner=mixed)
    term = random_terminal(spec, 9)
    with hjbdp.Backup(spec, variant=1) as bk:
        assert bk.info()["kernel_variant"] == 1
        out = bk.solve(4, terminal=term, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec, variant=0) as bk:
        assert bk.info()["kernel_variant"] == 0
        out0 = bk.solve(4, terminal=term, keep_J=True, keep_idx=True)
    ref = c_oracle.sweep(_abi, spec, 4, terminal=term, keep_J=True, keep_idx=True)
    for o in (out, out0):
        assert np.array_equal(o["J_stages"], ref["J_stages"])
        assert np.array_equal(o["idx_stages"], ref["idx_stages"])


ROWWISE = [
    # n, m, dtype, nonuniform, spread
    ((70, 8), (3,), np.float64, False, 0.3),
    ((130, 5, 4), (5,), np.float32, True, 0.4),
    ((9, 8, 7), (3, 2), np.float32, False, 0.2),
    ((66, 5, 4, 5), (9,), np.float32, True, 0.5),
    ((6, 5, 4, 5), (3, 4), np.float64, False, 0.3),
    ((4, 3, 4, 3, 5), (2, 3, 2), np.float32, False, 0.3),
    ((5, 3, 4, 3, 3, 4), (4,), np.float32, True, 0.2),
]


def _row_problem(n, m, dtype, nonuniform, spread):
    """nested_problem with the coupling to state dim 0 removed from axes >= 1 (variant 6's applicability rule)."""
    import hjbdp
    from problems import nested_problem
    spec = nested_problem(300 + len(n), n, m, dtype=dtype, nonuniform=nonuniform, spread=spread)
    nxt = [spec.next_terms[0]] + [[t for t in spec.next_terms[a] if 0 not in t.dims] for a in range(1, spec.D)]
    return hjbdp.ProblemSpec(spec.knots, spec.m, nxt, spec.cost_terms, dtype=dtype, index_base=1)


@pytest.mark.parametrize("n,m,dtype,nonuniform,spread", ROWWISE)
def test_rowwise_variant_bit_exact(env, n, m, dtype, nonuniform, spread):
    """Variant 6 (one wavefront per grid row, scalar cells/weights for axes 1..D-1)."""
    hjbdp, _abi, c_oracle = env
    from problems import random_terminal
    spec = _row_problem(n, m, dtype, nonuniform, spread)
    term = random_terminal(spec, 5)
    ref = c_oracle.sweep(_abi, spec, 3, terminal=term, keep_J=True, keep_idx=True)
    for lean in (1, 0):                       # the lean form (per-control records prepared lane-parallel) and the plain one
        with hjbdp.Backup(spec, variant=6) as bk:
            assert bk.info()["kernel_variant"] == 6
            bk.set_option("row_lean", lean)
            out = bk.solve(3, terminal=term, keep_J=True, keep_idx=True)
        assert np.array_equal(out["J_stages"], ref["J_stages"]), lean
        assert np.array_equal(out["idx_stages"], ref["idx_stages"]), lean


def test_rowwise_variant_pos_att_slab_and_f16(env):
    """pos-att is the shape variant 6 exists for (chosen automatically at C4 size, see test_gpu_solvers); here
    forced on a small grid: whole grid, a slab with halos, float16 J storage; refused for Kirk (x2+ depends on x1)."""
    hjbdp, _abi, c_oracle = env
    from problems import random_terminal
    pa = hjbdp.Solver_pos_att()
    pa.n_mesh_x, pa.n_mesh_v, pa.n_mesh_t, pa.n_mesh_w = 70, 6, 5, 10
    pa.cost_mode = "exact"                  # the materialised single(double sum) table (a float64-cost problem runs on kernels 5 / 7 only)
    sx, sv, st, sw = pa.grids()
    spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7,
                                    pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    with hjbdp.Backup(_kirk(hjbdp, "double", 5, 8, 9).build_spec()) as bk:
        with pytest.raises(hjbdp.HjbError):
            bk.set_option("variant", 6)
    term = random_terminal(spec, 3)
    ref = c_oracle.sweep(_abi, spec, 4, terminal=term)
    with hjbdp.Backup(spec, variant=6) as bk:
        need = bk.info()
        assert need["kernel_variant"] == 6
        out = bk.solve(4, terminal=term)
    assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
    b, e = 3, 7
    hl, hh = min(need["halo_needed_lo"], b), min(need["halo_needed_hi"], spec.n[-1] - e)
    inner = spec.nS // spec.n[-1]
    sub = np.asfortranarray(term.reshape(inner, -1, order="F")[:, b - hl:e + hh]).reshape(-1, order="F")
    Jr, ir = c_oracle.backup_stage(_abi, spec, sub, slab=(b, e, hl, hh))
    with hjbdp.Backup(spec, slab=(b, e, hl, hh), variant=6) as bk:
        assert bk.info()["kernel_variant"] == 6
        Jg, ig = bk.backup_stage(sub)
    assert np.array_equal(Jg, Jr) and np.array_equal(ig, ir)
    hspec = hjbdp.ProblemSpec(spec.knots, spec.m, spec.next_terms, spec.cost_terms, dtype=np.float32, index_base=1,
                              j_storage=np.float16)
    with hjbdp.Backup(hspec, variant=6) as bk:
        assert bk.info()["kernel_variant"] == 6
        oh = bk.solve(4)
    rh = c_oracle.sweep(_abi, hspec, 4)
    assert np.array_equal(oh["J"], rh["J"]) and np.array_equal(oh["idx"], rh["idx"])


_COLSWEEP_FORMS = set()
COLSWEEP = [
    # n, nU, nonuniform, gax, cost, a1_amp, levels, j_storage, terminal, dpp (a guess only: recorded, not asserted)
    ((70, 9, 8, 11), 9, False, 3, "fast", 0.6, 5, None, True, True),       # two chunks of axis 0
    ((130, 7, 5, 6), 9, False, 3, "fast", 0.6, 5, None, False, True),      # three chunks; zero terminal: every first-stage total ties
    ((64, 7, 9, 12), 9, True, 3, "fast", 0.6, 5, None, False, False),      # uneven knots: both axis-0 neighbours loaded
    ((33, 10, 12, 7), 9, True, 2, "fast", 0.5, 5, None, True, False),      # group axis = axis 2, window = the last axis
    ((63, 10, 12, 7), 9, False, 2, "fast", 0.5, 5, None, True, True),
    ((5, 4, 3, 4), 6, False, 3, "fast", 0.6, 3, None, True, True),         # tiny
    ((40, 13, 6, 9), 9, False, 3, "fast", 1.8, 5, None, True, True),       # axis-1 cells jump: re-priming inside a column
    ((37, 6, 7, 10), 12, False, 3, "multi", 0.6, 4, None, True, True),     # two control-only cost terms
    ((37, 6, 7, 10), 9, True, 2, "step01", 0.6, 5, None, True, False),     # per-step cost term over (dim 0, dim 1)
    ((20, 6, 7, 10), 7, False, 3, "ctrl_only", 0.6, 6, None, True, True),  # no state cost term; 6 groups
    ((66, 8, 9, 10), 9, False, 3, "fast", 0.6, 5, "f16", True, True),      # float16 cost-to-go storage
    ((66, 8, 9, 10), 9, True, 3, "fast", 0.6, 5, "f16", True, False),
    ((30, 8, 9, 10), 16, False, 3, "fast", 0.6, 5, None, True, True),      # 16 controls: groups split when their slots are full
]


@pytest.mark.parametrize("n,nU,nonuniform,gax,cost,a1_amp,levels,j_storage,terminal,dpp", COLSWEEP)
def test_colsweep_variant_bit_exact(env, n, nU, nonuniform, gax, cost, a1_amp, levels, j_storage, terminal, dpp):
    """Variant 7 (column sweep, kernels_colsweep.h): the two control-independent lerps once per corner row, rolling
    along axis 1; controls grouped by the cell of the group axis.  Bit-exact J and argmin against the oracle."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, random_terminal
    spec = colsweep_problem(700 + n[0] + nU, n, nU=nU, nonuniform=nonuniform, gax=gax, cost=cost, a1_amp=a1_amp,
                            levels=levels, j_storage=np.float16 if j_storage else None)
    term = random_terminal(spec, 11) if terminal else None
    ref = c_oracle.sweep(_abi, spec, 3, terminal=term, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec, variant=7) as bk:
        assert bk.info()["kernel_variant"] == 7
        dpp_on = bk.get_option("cs_dpp")                    # uneven knots usually rule the DPP form out (host check)
        _COLSWEEP_FORMS.add((dpp_on, dpp))
        assert bk.get_option("cs_group_axis") in (2, 3) and 1 <= bk.get_option("cs_groups") <= 6
        out = bk.solve(3, terminal=term, keep_J=True, keep_idx=True)
        assert np.array_equal(out["J_stages"], ref["J_stages"])
        assert np.array_equal(out["idx_stages"], ref["idx_stages"])
        for mod in (1, 3):                                   # any column -> XCD assignment gives the same result
            bk.set_option("cs_xcd_mod", mod)
            o2 = bk.solve(3, terminal=term)
            assert np.array_equal(o2["J"], ref["J"]) and np.array_equal(o2["idx"], ref["idx"]), mod
        bk.set_option("cs_xcd_mod", 0)
        auto_split = bk.get_option("cs_split")
        assert auto_split >= 1
        for parts in (1, 3, 8):                              # a column swept in several parts, each priming where it starts
            bk.set_option("cs_split", parts)
            assert bk.get_option("cs_split") == min(parts, n[1])
            o5 = bk.solve(3, terminal=term)
            assert np.array_equal(o5["J"], ref["J"]) and np.array_equal(o5["idx"], ref["idx"]), parts
        bk.set_option("cs_split", 0)
        assert bk.get_option("cs_split") == auto_split
        bk.set_option("cs_xcd_axis", 1)                      # the XCDs split the window axis instead of the group axis
        o4 = bk.solve(3, terminal=term)
        assert np.array_equal(o4["J"], ref["J"]) and np.array_equal(o4["idx"], ref["idx"])
        bk.set_option("cs_xcd_axis", 0)
        if dpp_on:                                           # the two-loads-per-row form on the same problem
            bk.set_option("cs_dpp", 0)
            assert bk.get_option("cs_dpp") == 0
            o3 = bk.solve(3, terminal=term)
            assert np.array_equal(o3["J"], ref["J"]) and np.array_equal(o3["idx"], ref["idx"])


COLCOOP = [
    # n, nU, nonuniform, gax, a1_amp, j_storage, terminal     (n0 a multiple of 4 - of 8 for float16 storage)
    ((64, 9, 8, 19), 9, False, 3, 0.6, None, True),         # three blocks of columns, the last one 3 wide
    ((72, 7, 11, 6), 9, False, 2, 0.5, None, True),         # group axis = axis 2; two chunks of axis 0; one partial block
    ((136, 6, 5, 9), 9, False, 3, 0.6, None, False),        # three chunks; zero terminal: every first-stage total ties
    ((64, 8, 9, 12), 9, True, 3, 0.6, None, True),          # uneven knots: any axis-0 cell pattern is fine here
    ((40, 13, 6, 9), 9, False, 3, 1.8, None, True),         # axis-1 cells jump: the workgroup re-primes together
    ((64, 8, 9, 10), 9, False, 3, 0.6, "f16", True),        # float16 cost-to-go storage (8 knots per staging load)
    ((8, 4, 3, 4), 6, False, 3, 0.6, None, True),           # tiny: one workgroup per group-axis index
]
_COLCOOP_RAN = []


@pytest.mark.parametrize("n,nU,nonuniform,gax,a1_amp,j_storage,terminal", COLCOOP)
def test_colsweep_cooperative_form_bit_exact(env, n, nU, nonuniform, gax, a1_amp, j_storage, terminal):
    """Variant 7, cooperative form (kernels_colcoop.h, option cs_coop): eight neighbouring columns share their corner
    rows through LDS.  Same plans, same arithmetic: bit-exact J and argmin against the oracle, every stage."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, random_terminal
    for a1_axis in (gax, 5 - gax):          # axis 1 may see the group axis only - and the plan picks the group axis
        spec = colsweep_problem(900 + n[0] + nU, n, nU=nU, nonuniform=nonuniform, gax=gax, cost="fast", a1_amp=a1_amp,
                                levels=5 if nU == 9 else 3, j_storage=np.float16 if j_storage else None, a1_axis=a1_axis)
        with hjbdp.Backup(spec, variant=7) as bk:
            bk.set_option("cs_coop", 1)
            if bk.get_option("cs_coop") or bk.get_option("cs_coop_why") != 2:
                break
    term = random_terminal(spec, 13) if terminal else None
    ref = c_oracle.sweep(_abi, spec, 3, terminal=term, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec, variant=7) as bk:
        assert bk.info()["kernel_variant"] == 7 and bk.get_option("cs_coop") == 0      # off unless asked for
        bk.set_option("cs_coop", 1)
        on = bk.get_option("cs_coop")                        # the form in effect: the host check may rule it out
        _COLCOOP_RAN.append((on, bk.get_option("cs_coop_why")))
        out = bk.solve(3, terminal=term, keep_J=True, keep_idx=True)
        assert np.array_equal(out["J_stages"], ref["J_stages"])
        assert np.array_equal(out["idx_stages"], ref["idx_stages"])
        bk.set_option("cs_coop", 0)
        assert bk.get_option("cs_coop") == 0


def test_colsweep_cooperative_form_was_exercised():
    """Runs after the parametrised cases: the host check must have admitted the cooperative form on most of them."""
    assert sum(on for on, _ in _COLCOOP_RAN) >= len(COLCOOP) - 2, _COLCOOP_RAN


def test_colsweep_both_forms_were_exercised():
    """Runs after the parametrised cases: the DPP form and the two-loads form must both have been selected by the
    host check on some case (otherwise the table above no longer covers one of them)."""
    forms = {f for f, _ in _COLSWEEP_FORMS}
    assert forms == {0, 1}, _COLSWEEP_FORMS


@pytest.mark.parametrize("gax", [2, 3])
def test_colsweep_variant_slab(env, gax):
    """Variant 7 on a slab of the last axis with halos (the multi-GPU form): gax = 2 shards the window axis (halo of
    one or two planes), gax = 3 the group axis (a halo as wide as the largest control displacement)."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, random_terminal
    spec = colsweep_problem(41 + gax, (36, 7, 9, 14) if gax == 3 else (36, 7, 14, 9), gax=gax, nonuniform=True)
    term = random_terminal(spec, 2)
    with hjbdp.Backup(spec, variant=7) as bk:
        need = bk.info()
    b, e = 4, 8 if gax == 3 else 7
    hl, hh = min(need["halo_needed_lo"], b), min(need["halo_needed_hi"], spec.n[-1] - e)
    inner = spec.nS // spec.n[-1]
    sub = np.asfortranarray(term.reshape(inner, -1, order="F")[:, b - hl:e + hh]).reshape(-1, order="F")
    Jr, ir = c_oracle.backup_stage(_abi, spec, sub, slab=(b, e, hl, hh))
    with hjbdp.Backup(spec, slab=(b, e, hl, hh), variant=7) as bk:
        assert bk.info()["kernel_variant"] == 7
        Jg, ig = bk.backup_stage(sub)
    assert np.array_equal(Jg, Jr) and np.array_equal(ig, ir)
    # no halo at all: reported, never silent
    sub2 = np.asfortranarray(term.reshape(inner, -1, order="F")[:, b:e]).reshape(-1, order="F")
    with hjbdp.Backup(spec, slab=(b, e, 0, 0), variant=7) as bk:
        with pytest.raises(hjbdp.HjbError) as ei:
            bk.backup_stage(sub2)
        assert ei.value.status == _abi.HJB_E_HALO


def test_colsweep_variant_refused_when_not_applicable(env):
    """pos-att in the reference's axis order (x, v, theta, w): axis 1 (v) moves with the control -> not variant 7's shape."""
    hjbdp, _abi, c_oracle = env
    pa = hjbdp.Solver_pos_att()
    pa.cost_mode = "terms"
    sx, sv, st, sw = pa.grids()
    spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7,
                                    pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
    with hjbdp.Backup(spec) as bk:
        with pytest.raises(hjbdp.HjbError):
            bk.set_option("variant", 7)
    # relabelled (x, theta, v, w) it applies, and agrees with the oracle on the relabelled problem bit for bit
    pspec, to_old = hjbdp.permute_state_axes(spec, (0, 2, 1, 3))
    ref = c_oracle.sweep(_abi, pspec, 4)
    with hjbdp.Backup(pspec, variant=7) as bk:
        assert bk.info()["kernel_variant"] == 7
        out = bk.solve(4)
    assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
    # and with the reference order to rounding (the lerp order differs: a few ulp), same policy almost everywhere
    ref0 = c_oracle.sweep(_abi, spec, 4)
    J_old = to_old(out["J"])
    assert np.max(np.abs(J_old - ref0["J"])) <= 2e-6 * np.max(np.abs(ref0["J"]))
    assert np.mean(to_old(out["idx"]) == ref0["idx"]) > 0.999


PACKED = [
    ((9, 8), (3,), False, "inc"),
    ((40, 37), (7,), True, "dec"),                 # > 512 states, odd tail
    ((9, 8, 7), (5, 4, 3), False, "inc"),          # C2 shape, 504 states (< one workgroup pass)
    ((13, 11, 9), (4, 5), True, "dec"),
    ((6, 5, 4, 5), (3, 4), False, "inc"),
    ((4, 3, 4, 3, 3, 5), (3, 3, 3), False, "dec"),  # C3 shape
]


@pytest.mark.parametrize("n,m,nonuniform,mono", PACKED)
def test_packed_variant_bit_exact(env, n, m, nonuniform, mono):
    """Variant 2 (two states per lane, packed fp32, one-sided cell test) is chosen
    for float32 spacecraft-shaped problems with a monotone inner control table and
    agrees bit for bit with the oracle; large spread forces many cell crossings."""
    hjbdp, _abi, c_oracle = env
    from problems import nested_problem, random_terminal
    spec = nested_problem(777 + len(n), n, m, dtype=np.float32, nonuniform=nonuniform, monotone=mono, spread=0.45)
    term = random_terminal(spec, 11)
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["kernel_variant"] in (2, 4)
        out = bk.solve(4, terminal=term, keep_J=True, keep_idx=True)
    ref = c_oracle.sweep(_abi, spec, 4, terminal=term, keep_J=True, keep_idx=True)
    assert np.array_equal(out["J_stages"], ref["J_stages"])
    assert np.array_equal(out["idx_stages"], ref["idx_stages"])
    for v in (0, 1, 2, 4):
        with hjbdp.Backup(spec, variant=v) as bk:
            assert bk.info()["kernel_variant"] == v
            o = bk.solve(4, terminal=term)
        assert np.array_equal(o["J"], ref["J"]) and np.array_equal(o["idx"], ref["idx"]), v


@pytest.mark.parametrize("n,m", [((9, 8), (3,)), ((9, 8, 7), (4, 5, 3)), ((4, 3, 4, 3, 3, 5), (3, 3, 3))])
def test_packed_variant_state_dependent_inner_term(env, n, m):
    """Variant 4 also takes a last-axis inner term that depends on the state (attitude:
    h*((J1-J2)/J3*w1*w2 + u3/J3), Solver_attitude.m:425); variant 2 does not."""
    hjbdp, _abi, c_oracle = env
    from problems import nested_problem, random_terminal
    spec = nested_problem(99 + len(n), n, m, dtype=np.float32, mixed_inner="only", spread=0.4)
    term = random_terminal(spec, 5)
    ref = c_oracle.sweep(_abi, spec, 4, terminal=term)
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["kernel_variant"] == 4
        o = bk.solve(4, terminal=term)
        with pytest.raises(hjbdp.HjbError):
            bk.set_option("variant", 2)
    assert np.array_equal(o["J"], ref["J"]) and np.array_equal(o["idx"], ref["idx"])


def test_packed_variant_slab(env):
    hjbdp, _abi, c_oracle = env
    from problems import nested_problem, random_terminal
    spec = nested_problem(55, (9, 8, 14), (4, 3), dtype=np.float32, monotone="inc", spread=0.1)
    term = random_terminal(spec, 3)
    Jw, iw = c_oracle.backup_stage(_abi, spec, term)
    T3 = term.reshape(72, 14, order="F")
    b, e, lo, hi = 5, 10, 2, 2
    for v in (2, 4, 5):
        with hjbdp.Backup(spec, slab=(b, e, lo, hi), variant=v) as bk:
            assert bk.info()["kernel_variant"] == v
            Jo, io = bk.backup_stage(np.asfortranarray(T3[:, b - lo:e + hi]).reshape(-1, order="F"))
        assert np.array_equal(Jo.reshape(72, -1, order="F")[:, lo:lo + e - b], Jw.reshape(72, 14, order="F")[:, b:e])
        assert np.array_equal(io, iw.reshape(72, 14, order="F")[:, b:e].reshape(-1, order="F"))


def test_packed_modes_2_and_3_slab(env):
    """Slabs (multi-GPU decomposition of the last axis) through variant 4's state-window modes: a 5-D nested
    problem (mode 2) and the attitude problem with the on-the-fly quaternion model (mode 3); the owned planes of
    the slab must equal the same planes of the whole-grid oracle result."""
    hjbdp, _abi, c_oracle = env
    from problems import nested_problem, random_terminal
    spec = nested_problem(91, (4, 3, 5, 4, 12), (3, 4, 3), dtype=np.float32, monotone="inc", spread=0.15)
    sa = hjbdp.Solver_attitude(n_mesh_w=9, n_mesh_q=4)
    sa.U_vector = np.linspace(-0.11, 0.11, 5)
    for sp in (spec, sa.build_spec_model()):
        nl = sp.n[-1]
        inner = sp.nS // nl
        term = random_terminal(sp, 7)
        Jw, iw = c_oracle.backup_stage(_abi, sp, term)
        T2 = term.reshape(inner, nl, order="F")
        with hjbdp.Backup(sp) as bk:
            need = bk.info()
            assert need["kernel_variant"] == 4
        b, e = 3, 7
        lo, hi = min(need["halo_needed_lo"], b), min(need["halo_needed_hi"], nl - e)
        with hjbdp.Backup(sp, slab=(b, e, lo, hi)) as bk:
            assert bk.info()["kernel_variant"] == 4
            Jo, io = bk.backup_stage(np.asfortranarray(T2[:, b - lo:e + hi]).reshape(-1, order="F"))
        assert np.array_equal(Jo.reshape(inner, -1, order="F")[:, lo:lo + e - b], Jw.reshape(inner, nl, order="F")[:, b:e])
        assert np.array_equal(io, iw.reshape(inner, nl, order="F")[:, b:e].reshape(-1, order="F"))
        # too small a halo is reported, never silently wrong
        if lo > 0:
            with hjbdp.Backup(sp, slab=(b, e, lo - 1, hi)) as bk:
                with pytest.raises(hjbdp.HjbError) as ei:
                    bk.backup_stage(np.asfortranarray(T2[:, b - lo + 1:e + hi]).reshape(-1, order="F"))
                assert ei.value.status == _abi.HJB_E_HALO


EDGE = [
    # (n, m, dtype): minimum axis sizes, single-control dims, inner dim of 1/2/odd size, nU == 64
    ((2, 2), (1,), np.float32),
    ((2, 3, 2), (1, 1, 1), np.float32),
    ((5, 4, 2), (3, 1, 2), np.float32),
    ((5, 4, 3), (1, 2, 1), np.float32),
    ((6, 5), (64,), np.float32),
    ((6, 5), (65,), np.float64),
    ((3, 3, 3, 3), (2, 32), np.float32),      # inner dim at the packed kernels' 32-control limit
    ((3, 3, 3), (2, 33), np.float32),         # ... and just above it (falls back to variant 1)
    ((300, 2), (5,), np.float32),
]


@pytest.mark.parametrize("n,m,dtype", EDGE)
def test_edge_shapes_all_variants(env, n, m, dtype):
    hjbdp, _abi, c_oracle = env
    from problems import nested_problem, random_terminal
    spec = nested_problem(2024, n, m, dtype=dtype, spread=0.5)
    term = random_terminal(spec, 1)
    ref = c_oracle.sweep(_abi, spec, 3, terminal=term)
    seen = set()
    for v in (None, 0, 1, 2, 3, 4, 5, 6):
        try:
            bk = hjbdp.Backup(spec, variant=v)
        except hjbdp.HjbError as e:
            assert e.status == _abi.HJB_E_UNSUPPORTED and v in (1, 2, 4, 6)
            continue
        with bk:
            seen.add(bk.info()["kernel_variant"])
            o = bk.solve(3, terminal=term)
        assert np.array_equal(o["J"], ref["J"]) and np.array_equal(o["idx"], ref["idx"]), (v, n, m)
    assert {0, 3} <= seen


@pytest.mark.order(2)
def test_c2_workload_small_bit_exact(env):
    """BASELINE configs[1] (Solver_position 3-DOF) at a size the oracle finishes in seconds."""
    hjbdp, _abi, c_oracle = env
    from hjbdp.synthetic import position3d_spec
    spec = position3d_spec(n=15, mu=7)
    ref = c_oracle.sweep(_abi, spec, 6, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["kernel_variant"] == 4
        # the C2 shape runs WITHOUT an axis-0 (cell, t) table: the kernel forms the entry from q in registers (K3 mode 4) ...
        assert bk.get_option("axis0_table") == 0
        out = bk.solve(6, keep_J=True, keep_idx=True)
        assert np.array_equal(out["J_stages"], ref["J_stages"])
        assert np.array_equal(out["idx_stages"], ref["idx_stages"])
        # ... and with the table built after all (mode 1): the same bits
        bk.set_option("axis0_table", 1)
        assert bk.get_option("axis0_table") == 1
        out = bk.solve(6, keep_J=True, keep_idx=True)
        assert np.array_equal(out["J_stages"], ref["J_stages"])
        assert np.array_equal(out["idx_stages"], ref["idx_stages"])
        bk.set_option("variant", 2)                       # the two-states-per-lane kernel reads every axis from its table
        out = bk.solve(6)
        assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
    with hjbdp.Backup(spec, variant=2) as bk:             # forced straight away: the table is built on demand
        assert bk.get_option("axis0_table") == 1
        out = bk.solve(6)
        assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
    for n_last, mu in ((9, 5), (12, 4)):                  # other sizes of the last axis and of the control grid
        sp = position3d_spec(n=11, mu=mu, n_last=n_last)
        r2 = c_oracle.sweep(_abi, sp, 3)
        with hjbdp.Backup(sp) as bk:
            assert bk.get_option("axis0_table") == 0
            o2 = bk.solve(3)
        assert np.array_equal(o2["J"], r2["J"]) and np.array_equal(o2["idx"], r2["idx"])
    with hjbdp.MultiBackup(spec, [0, 0, 0]) as mb:       # slabs: every slab handle runs the table-less mode
        om = mb.solve(6)
    assert np.array_equal(om["J"], ref["J"]) and np.array_equal(om["idx"], ref["idx"])


def _chain_spec(n, m, gain, u_lo, u_hi, cost_order, h=0.2):
    """D-axis chain x+ = (I + h N) x + gain_a * u_a on the last three axes (N strictly upper-triangular ones), float32:
    the shape variant 4 contracts hierarchically (D = 3: kernels_packed2.h mode 4 / 1, D >= 4: the window mode 2).
    gain / the control range place the last axis' cell change on any control of the sweep, on none, or on several;
    cost_order picks the cost terms (without state terms a control term is the first of the canonical sum)."""
    from hjbdp.problem import ProblemSpec, Term
    D = len(n)
    f = np.float32
    knots = [np.linspace(-1.0, 1.0, k).astype(np.float32) for k in n]
    A = np.eye(D) + h * np.triu(np.ones((D, D)), 1)
    nxt = []
    for a in range(D):
        terms = [Term((j,), f(A[a, j]) * knots[j]) for j in range(D) if A[a, j] != 0.0]
        c = a - (D - 3)
        if c >= 0:
            u = np.linspace(u_lo, u_hi, m[c]) if m[c] > 1 else np.array([0.5 * (u_lo + u_hi)])
            terms.append(Term((D + c,), (gain[c] * u).astype(np.float32)))
        nxt.append(terms)
    cost = [Term((j,), f(1.5 + j) * knots[j] ** 2) for j in range(D)]
    for c in range(3):
        u = np.linspace(u_lo, u_hi, m[c]) if m[c] > 1 else np.array([0.5 * (u_lo + u_hi)])
        cost.append(Term((D + c,), (f(0.3 + 0.1 * c) * u ** 2).astype(np.float32)))
    cost = [cost[i] for i in cost_order(len(cost))]
    return ProblemSpec(knots, list(m), nxt, cost, dtype=np.float32, index_base=1)


K3_TRIPS = [
    # n, m (o0, o1, inner), gain per control, control range, cost order
    ((9, 8, 7), (3, 4, 6), (0.05, 0.05, 0.10), (-1.0, 1.0), "std"),        # even sweep, change inside
    ((9, 8, 7), (3, 5, 7), (0.05, 0.05, 0.10), (-1.0, 1.0), "std"),        # odd sweep, odd step count (a single step at the end)
    ((9, 8, 7), (2, 2, 5), (0.05, 0.05, 0.02), (0.2, 1.0), "std"),         # no cell change in the sweep
    ((9, 8, 7), (4, 6, 5), (0.05, 0.05, 0.30), (-1.0, 1.0), "std"),        # change on the last control of an odd sweep / other parities by state
    ((9, 8, 7), (3, 4, 8), (0.05, 0.05, 0.90), (-1.0, 1.0), "std"),        # several changes per sweep: the one-step path
    ((9, 8, 7), (3, 4, 1), (0.05, 0.05, 0.10), (-1.0, 1.0), "std"),        # one inner control
    ((9, 8, 7), (3, 1, 5), (0.05, 0.05, 0.10), (-1.0, 1.0), "std"),        # one level-1 step: no trip at all
    ((9, 8, 7), (3, 6, 5), (0.05, 0.40, 0.10), (-1.0, 1.0), "std"),        # level-1 axis moves two cells: rows re-selected, window left
    ((9, 8, 7), (3, 6, 5), (0.05, 0.05, 0.10), (-1.0, 1.0), "l1_first"),   # the level-1 cost term leads the sum
    ((9, 8, 7), (3, 6, 4), (0.05, 0.05, 0.10), (-1.0, 1.0), "l0_first"),   # the level-0 cost term leads the sum
    ((9, 8, 7), (3, 6, 5), (0.05, 0.05, 0.10), (-1.0, 1.0), "inner_first"),
    ((5, 6, 7, 6), (3, 4, 5), (0.05, 0.05, 0.10), (-1.0, 1.0), "std"),     # D = 4: the window mode
    ((5, 6, 7, 6), (2, 5, 6), (0.30, 0.40, 0.30), (-1.0, 1.0), "l1_first"),
    ((3, 4, 5, 6, 5), (3, 3, 4), (0.05, 0.10, 0.20), (-0.5, 1.0), "std"),  # D = 5
    ((70, 6, 5), (3, 5, 7), (0.05, 0.05, 0.10), (-1.0, 1.0), "std"),       # waves inside one axis-0 row: the wave-uniform trip
    ((70, 6, 5), (3, 6, 6), (0.05, 0.05, 0.10), (-1.0, 0.6), "l1_first"),
    ((66, 5, 4, 5), (3, 5, 5), (0.05, 0.05, 0.10), (-1.0, 1.0), "std"),
    ((66, 5, 4, 5), (3, 4, 6), (0.05, 0.05, 0.12), (-0.7, 1.0), "inner_first"),
]


@pytest.mark.parametrize("n,m,gain,urange,order", K3_TRIPS)
def test_packed2_trip_shapes_bit_exact(env, n, m, gain, urange, order):
    """Variant 4's two-step trip (kernels_packed2.h): sweeps of odd and even length, the wave-wide cell change on either
    control of a pair / on the last control / absent / repeated, kept lerp rows re-selected, a control cost term first in
    the canonical sum - labels and values against the oracle, every stage."""
    hjbdp, _abi, c_oracle = env
    from problems import random_terminal
    D = len(n)
    orders = {                                   # which cost terms, in the canonical order (state terms, then one per control)
        "std": lambda k: list(range(k)),
        "l0_first": lambda k: [D, D + 1, D + 2],     # no state terms: the level-0 control term leads the sum
        "l1_first": lambda k: [D + 1, D + 2],        # ... the level-1 term leads it
        "inner_first": lambda k: [0, D + 2],         # no level terms at all
    }
    spec = _chain_spec(n, m, gain, urange[0], urange[1], orders[order])
    term = random_terminal(spec, 5)
    ref = c_oracle.sweep(_abi, spec, 3, terminal=term, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec, variant=4) as bk:
        assert bk.info()["kernel_variant"] == 4
        mode = bk.get_option("packed2_mode")
        assert mode == 4 if D == 3 else mode in (2, 5), mode      # the hierarchical modes, not the plain one
        out = bk.solve(3, terminal=term, keep_J=True, keep_idx=True)
        assert np.array_equal(out["J_stages"], ref["J_stages"]), (n, m, order, mode)
        assert np.array_equal(out["idx_stages"], ref["idx_stages"]), (n, m, order, mode)
        if mode == 5:                                         # the four-plane form of the same window: the same bits
            bk.set_option("window_planes", 4)
            assert bk.get_option("packed2_mode") == 2
            out = bk.solve(3, terminal=term, keep_J=True, keep_idx=True)
    assert np.array_equal(out["J_stages"], ref["J_stages"]), (n, m, order, mode)
    assert np.array_equal(out["idx_stages"], ref["idx_stages"]), (n, m, order, mode)


@pytest.mark.parametrize("gain,near", [((0.05, 0.10, 0.12), True), ((0.30, 0.40, 0.90), False)])
def test_packed2_window_planes(env, gain, near):
    """D = 6: when the inner control moves the last axis by less than a cell per step the per-state window holds three
    last-axis planes (modes 5 / 6: four workgroups per CU on the attitude grids), else four; option window_planes switches a
    qualifying handle between the two forms - the same bits either way."""
    hjbdp, _abi, c_oracle = env
    from problems import random_terminal
    spec = _chain_spec((20, 3, 4, 5, 4, 6), (3, 5, 5), gain, -1.0, 1.0, lambda k: list(range(k)))
    term = random_terminal(spec, 9)
    ref = c_oracle.sweep(_abi, spec, 2, terminal=term, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec, variant=4) as bk:
        assert bk.get_option("packed2_mode") == (5 if near else 2)
        forms = (3, 4, 3) if near else (4,)
        for planes in forms:
            bk.set_option("window_planes", planes)
            assert bk.get_option("packed2_mode") == (5 if planes == 3 else 2)
            out = bk.solve(2, terminal=term, keep_J=True, keep_idx=True)
            assert np.array_equal(out["J_stages"], ref["J_stages"]), planes
            assert np.array_equal(out["idx_stages"], ref["idx_stages"]), planes
        if not near:
            with pytest.raises(hjbdp.HjbError) as ei:
                bk.set_option("window_planes", 3)
            assert ei.value.status == _abi.HJB_E_UNSUPPORTED


def test_variant_1_refused_when_not_applicable(env):
    hjbdp, _abi, c_oracle = env
    spec = _kirk(hjbdp, "double", 5, 8, 9).build_spec()   # both axes depend on u
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["kernel_variant"] == 5           # general shape -> table-driven generic kernel
        with pytest.raises(hjbdp.HjbError) as ei:
            bk.set_option("variant", 1)
        assert ei.value.status == _abi.HJB_E_UNSUPPORTED


def test_ctrlsplit_variant_selected_for_kirk(env):
    """Few states x many controls -> variant 3 (wave min-reduction over the control axis),
    J staged in LDS; ties across lanes keep the first index."""
    hjbdp, _abi, c_oracle = env
    spec = _kirk(hjbdp, "single", 6, 40, 300).build_spec()
    with hjbdp.Backup(spec) as bk:
        inf = bk.info()
        assert inf["kernel_variant"] == 3 and inf["lds_bytes"] == 40 * 40 * 4
        out = bk.solve(5, keep_J=True, keep_idx=True)
    ref = c_oracle.sweep(_abi, spec, 5, keep_J=True, keep_idx=True)
    assert np.array_equal(out["J_stages"], ref["J_stages"]) and np.array_equal(out["idx_stages"], ref["idx_stages"])
    # every control ties (cost and next state independent of u): label must be the first
    k = np.linspace(-1, 1, 7)
    tie = hjbdp.ProblemSpec([k, k], [200], [[hjbdp.Term((0,), k)], [hjbdp.Term((1,), k)]],
                            [hjbdp.Term((0,), k ** 2), hjbdp.Term((2,), np.zeros(200))], dtype=np.float64, index_base=1)
    with hjbdp.Backup(tie, variant=3) as bk:
        J, idx = bk.backup_stage(np.arange(49, dtype=np.float64))
    assert np.all(idx == 1)


def test_exact_ties_first_index_wins(env):
    """MATLAB min returns the first index among equal minima; cascade order for C=2."""
    hjbdp, _abi, c_oracle = env
    k = np.linspace(-1, 1, 6)
    # next state and cost independent of the control -> every control ties
    spec = hjbdp.ProblemSpec([k, k], [4, 3], [[hjbdp.Term((0,), k)], [hjbdp.Term((1,), k)]],
                             [hjbdp.Term((0,), k ** 2), hjbdp.Term((2,), np.zeros(4)), hjbdp.Term((3,), np.zeros(3))],
                             dtype=np.float32, index_base=1)
    with hjbdp.Backup(spec) as bk:
        J, idx = bk.backup_stage(np.zeros(36, np.float32))
    assert np.all(idx == 1)
    # controls (i1=2,i2=0) and (i1=0,i2=1) (0-based) tie for the minimum: the cascade prefers i1=0
    M = np.full((4, 3), 5.0)
    M[0, 1] = M[2, 0] = -1.0
    spec2 = hjbdp.ProblemSpec([k, k], [4, 3], [[hjbdp.Term((0,), k)], [hjbdp.Term((1,), k)]],
                              [hjbdp.Term((0,), k ** 2), hjbdp.Term((2, 3), M)],
                              dtype=np.float32, index_base=0)
    with hjbdp.Backup(spec2) as bk:
        J, idx = bk.backup_stage(np.zeros(36, np.float32))
    assert np.all(idx == 0 + 4 * 1)   # label = i1 + m1*i2 with (i1,i2) = (0,1)


def test_slab_with_halo_matches_whole_grid(env):
    """Multi-GPU building block: a slab handle with halos reproduces its part of
    the whole-grid backup bit for bit."""
    hjbdp, _abi, c_oracle = env
    from problems import random_problem, random_terminal
    spec = random_problem(99, (9, 8, 12), (4, 3), dtype=np.float32, spread=0.08)
    term = random_terminal(spec, 3)
    with hjbdp.Backup(spec) as bk:
        Jw, iw = bk.backup_stage(term)
        need = bk.info()
    inner = 9 * 8
    Jw3 = Jw.reshape(inner, 12, order="F")
    iw3 = iw.reshape(inner, 12, order="F")
    T3 = term.reshape(inner, 12, order="F")
    hl, hh = need["halo_needed_lo"], need["halo_needed_hi"]
    assert hl + hh < 12
    for (b, e) in [(0, 5), (5, 9), (9, 12)]:
        lo, hi = min(hl, b), min(hh, 12 - e)
        with hjbdp.Backup(spec, slab=(b, e, lo, hi)) as bk:
            Jin = np.asfortranarray(T3[:, b - lo:e + hi])
            Jo, io = bk.backup_stage(Jin.reshape(-1, order="F"))
        Jo = Jo.reshape(inner, e + hi - b + lo, order="F")
        assert np.array_equal(Jo[:, lo:lo + e - b], Jw3[:, b:e])
        assert np.array_equal(io.reshape(inner, e - b, order="F"), iw3[:, b:e])


def test_halo_violation_is_reported(env):
    hjbdp, _abi, c_oracle = env
    from problems import random_problem, random_terminal
    spec = random_problem(5, (6, 5, 16), (3,), dtype=np.float32, spread=0.6)
    term = random_terminal(spec, 1)
    with hjbdp.Backup(spec, slab=(6, 10, 0, 0)) as bk:
        with pytest.raises(hjbdp.HjbError) as ei:
            bk.backup_stage(term.reshape(30, 16, order="F")[:, 6:10].reshape(-1, order="F"))
        assert ei.value.status == _abi.HJB_E_HALO


def test_monitor_early_stop_matches_oracle(env):
    """Solver_pos_att.m:268-285 restated: stop when |d sum(J)| < tol at k_s % period == 0."""
    hjbdp, _abi, c_oracle = env
    from problems import random_problem
    spec = random_problem(21, (8, 7), (5,), dtype=np.float64, spread=0.05)
    events = []
    with hjbdp.Backup(spec) as bk:
        out = bk.solve(40, monitor_period=5, monitor_tol=1e9, progress=lambda *a: events.append(a))
    ref = c_oracle.sweep(_abi, spec, 40, monitor_period=5, monitor_tol=1e9)
    # tol huge: stops at the first monitor point, k_s = 40
    assert out["stopped_early"] and out["stages_done"] == ref["stages_done"] == 1
    assert np.array_equal(out["J"], ref["J"])
    assert events and events[0][0] == 40
    with hjbdp.Backup(spec) as bk:
        out = bk.solve(23, monitor_period=5, monitor_tol=0.0)
    ref = c_oracle.sweep(_abi, spec, 23, monitor_period=5, monitor_tol=0.0)
    assert out["stages_done"] == 23 and not out["stopped_early"]
    assert np.array_equal(out["J"], ref["J"])
    assert abs(out["last_e"] - ref["last_e"]) <= 1e-9 * max(1.0, abs(ref["last_e"]))
    assert out["last_e2"] == ref["last_e2"]


@pytest.mark.parametrize("monitor", [0, 40])
def test_graph_replay_matches_eager_and_oracle(env, monitor):
    """Long sweeps replay the ping-pong stage loop from a hipGraph (32 launches per
    replay); results, stage counts and the monitor must not change."""
    hjbdp, _abi, c_oracle = env
    from problems import nested_problem, random_terminal
    spec = nested_problem(8, (12, 11), (3,), dtype=np.float64, spread=0.05)
    term = random_terminal(spec, 4)
    n_st = 151                                     # odd: exercises the parity-fixing eager stage
    with hjbdp.Backup(spec) as bk:
        a = bk.solve(n_st, terminal=term, monitor_period=monitor, monitor_tol=0.0)
        a2 = bk.solve(n_st, terminal=term, monitor_period=monitor, monitor_tol=0.0)   # cached graph
        bk.set_option("graph", 0)
        b = bk.solve(n_st, terminal=term, monitor_period=monitor, monitor_tol=0.0)
    ref = c_oracle.sweep(_abi, spec, n_st, terminal=term, monitor_period=monitor, monitor_tol=0.0)
    for o in (a, a2, b):
        assert o["stages_done"] == n_st
        assert np.array_equal(o["J"], ref["J"]) and np.array_equal(o["idx"], ref["idx"])
        assert o["last_e2"] == ref["last_e2"]


_TORCH_COLD = []      # set when torch did not come up in a child: the torchrun tests after it do not wait again


@pytest.mark.order(95)
@pytest.mark.watchdog(1000)
def test_device_buffer_entry_point_with_torch(env):
    """hjb_backup_stage_device on torch-owned HBM buffers and a torch stream - in a CHILD interpreter
    (tests/torch_interop_child.py): the pytest process itself never loads torch's multi-gigabyte ROCm stack (a cold
    box pages it in from the image for minutes), and a child that does not come up within its limit is reported as
    such instead of eating the run's budget.  The child checks the result against the oracle itself."""
    import subprocess
    import sys
    from pathlib import Path
    child = Path(__file__).resolve().parent / "torch_interop_child.py"
    try:
        r = subprocess.run([sys.executable, str(child)], capture_output=True, text=True, timeout=900)
    except subprocess.TimeoutExpired:
        _TORCH_COLD.append(1)
        pytest.skip("torch did not come up within 900 s in a fresh interpreter on this box (cold image); "
                    "the interop path itself is unchanged and was not exercised")
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "TORCH_INTEROP_OK" in r.stdout


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_policy_lookup_bit_exact(env, dtype):
    """hjb_policy_lookup ('nearest' policy tables, 'linear' value/policy lookups) vs the C oracle."""
    hjbdp, _abi, c_oracle = env
    rng = np.random.default_rng(3)
    for D in (1, 2, 4):
        knots = [np.sort(rng.uniform(-1, 1, 6 + a)).astype(dtype).astype(np.float64) for a in range(D)]
        V = rng.standard_normal(tuple(len(k) for k in knots)).astype(dtype)
        pts = rng.uniform(-1.4, 1.4, size=(5000, D)).astype(dtype)
        pts[:50] = np.stack([rng.choice(k, 50) for k in knots], axis=1)      # exactly on knots
        for method in ("nearest", "linear"):
            got = hjbdp.policy_lookup(knots, V, pts, method)
            ref = c_oracle.lookup(_abi, knots, V, pts, method)
            assert np.array_equal(got, ref), (D, method)
    # the policy object of the position solver uses it for batched queries
    pol = hjbdp.solver_position.NearestPolicy([np.linspace(-1, 1, 5), np.linspace(-1, 1, 4)],
                                              rng.standard_normal((5, 4)))
    q = rng.uniform(-1, 1, size=(100, 2))
    assert np.array_equal(pol.lookup_many(q), np.array([pol(*p) for p in q]))


@pytest.mark.order(2)
def test_c2_full_size_properties(env):
    """BASELINE configs[1] at its FULL size (101^3 states x 21^3 controls): too big for a whole
    oracle sweep, so (1) two independent kernels (control-nested variant 1 and packed variant 4)
    must agree bit for bit on every state, (2) one whole plane of stage 2 is recomputed by the
    oracle from the GPU's own stage-1 J (slab with halos) and must match bit for bit, and
    (3) size-independent properties hold: J >= 0, J(0) = 0, J(x) = J(-x), J_k monotone in the horizon."""
    hjbdp, _abi, c_oracle = env
    from hjbdp.synthetic import position3d_spec
    spec = position3d_spec(n=101, mu=21)
    with hjbdp.Backup(spec) as bk:
        assert bk.info()["kernel_variant"] == 4
        o4 = bk.solve(2, keep_J=True, keep_idx=True)
    with hjbdp.Backup(spec, variant=1) as bk:
        o1 = bk.solve(2)
    assert np.array_equal(o4["J"], o1["J"]) and np.array_equal(o4["idx"], o1["idx"])
    J1 = o4["J_stages"][:, 1].reshape(101 * 101, 101, order="F")      # stage computed first (k_s = 2)
    J2 = o4["J_stages"][:, 0].reshape(101 * 101, 101, order="F")
    I2 = o4["idx_stages"][:, 0].reshape(101 * 101, 101, order="F")
    p = 37
    Jo, io = c_oracle.backup_stage(_abi, spec, np.asfortranarray(J1[:, p - 1:p + 2]).reshape(-1, order="F"),
                                   slab=(p, p + 1, 1, 1))
    assert np.array_equal(Jo.reshape(101 * 101, 3, order="F")[:, 1], J2[:, p])
    assert np.array_equal(io, I2[:, p])
    G = o4["J"].reshape(101, 101, 101, order="F")
    assert G.min() >= 0.0 and G[50, 50, 50] == 0.0
    assert np.allclose(G, G[::-1, ::-1, ::-1], rtol=2e-5, atol=1e-6)     # symmetric grid and control set
    assert np.all(J2 >= J1 - 1e-6)                                       # longer horizon never cheaper
    lab = o4["idx"] - 1
    u = np.stack([lab % 21, (lab // 21) % 21, lab // 441], axis=1)
    assert np.array_equal(u[(50 + 101 * (50 + 101 * 50))], [10, 10, 10])  # u*(0) = 0


@pytest.mark.order(96)
@pytest.mark.watchdog(1300)
@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_sharded_bench_matches_single_rank(env, overlap):
    """bench.py's multi-GPU path on ONE GPU: two torchrun ranks share cuda:0 and exchange halos over
    gloo (the RCCL transport itself needs >1 GPU); strong scaling = the same grid, so the summed checksum must
    equal a single-rank run.  Small pos-att grid, the column-sweep kernel forced (variant 7), with the
    interior / boundary-strip split (three slab handles per rank) and without."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    if _TORCH_COLD:
        pytest.skip("torch did not come up in a child process on this box (see test_device_buffer_entry_point_with_torch)")
    root = Path(__file__).resolve().parent.parent
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    common = ["--steps", "4", "--warmup", "1", "--grid-n", "34", "--variant", "7", "--no-cpu-baseline", "--no-pmc", "--no-extras"]
    try:
        one = subprocess.run([sys.executable, str(root / "bench.py")] + common, capture_output=True, text=True, timeout=600)
        assert one.returncode == 0, one.stderr[-2000:]
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), str(root / "bench.py"),
                              "--gpus", "2", "--backend", "gloo", "--share-gpu"] + common + ([] if overlap else ["--no-overlap"]),
                             capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        _TORCH_COLD.append(1)
        pytest.skip("a torch child process did not finish within 600 s on this box (cold image)")
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    b = json.loads(two.stdout.strip().splitlines()[-1])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and "17 of 34 planes per GPU, halo 0/1" in b["config"]["sharding"]
    assert ("overlapped" in b["config"]["sharding"]) == overlap
    assert a["config"]["kernel_variant"] == 7 and b["config"]["kernel_variant"] == 7
    assert abs(a["checksum_sum_J"] - b["checksum_sum_J"]) <= 1e-12 * abs(a["checksum_sum_J"])
    assert b["config"]["states_per_gpu"] * 2 == a["config"]["states_per_gpu"] == b["config"]["states"]
    # the line says which transport carried the halos and how many ranks its communicator reports (under RCCL it also times the
    # other transport - the RCCL calls inside libhjbdp - in the same run; two ranks on ONE device cannot form an RCCL communicator)
    tr = b["transports"]
    assert tr["headline"] == "torch" and tr["torch"]["comm_ranks"] == 2 and tr["checksums_equal"]
    assert tr["torch"]["checksum_sum_J"] == b["checksum_sum_J"] and "transports" not in a


@pytest.mark.order(96)
@pytest.mark.watchdog(700)
@pytest.mark.parametrize("who", [1, -2])
def test_two_rank_bench_survives_a_hang_after_the_headline(env, who):
    """bench.py at N > 1 measures its headline leg first; the parts behind it (the other halo transport's leg, the weak-scaling extra)
    have never met multi-GPU hardware.  If they hang - here: rank 1 alone, then every rank, never leaves them (a test hook) - every rank
    leaves on its own timer, rank 0 still prints ONE line: the headline-only one when it was itself stuck, the complete one when only
    its peer was (it waits at the final barrier with the complete line in hand), and the run ends with exit code 0."""
    import json
    import socket
    import subprocess
    import sys
    import time
    from pathlib import Path
    if _TORCH_COLD:
        pytest.skip("torch did not come up in a child process on this box")
    root = Path(__file__).resolve().parent.parent
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    common = ["--steps", "4", "--warmup", "1", "--grid-n", "34", "--variant", "7", "--no-cpu-baseline", "--no-pmc", "--no-extras"]
    t0 = time.time()
    try:
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), str(root / "bench.py"),
                              "--gpus", "2", "--backend", "gloo", "--share-gpu", "--test-hang-optional", str(who), "--optional-timeout", "12"] + common,
                             capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        pytest.fail("the hung optional part took the whole run with it")
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [ln for ln in two.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, two.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["metric"] == "bellman_backups_per_s" and b["value"] > 0 and b["roofline"]["frac"] > 0
    assert ("incomplete" in b) == (who == -2)            # rank 0 stuck itself: the headline-only line; only its peer: the complete one
    assert time.time() - t0 < 300


@pytest.mark.order(96)
@pytest.mark.watchdog(1300)
def test_two_rank_c3_bench_matches_single_rank(env):
    """`bench.py --gpus N --workload c3` (BASELINE configs[2] sharded along w3; 22 GB of J per rank at 51^6 on eight GPUs) on ONE GPU at
    16^6: two torchrun ranks share cuda:0 and exchange their 4 MB halo plane over gloo; the summed checksum equals the single-rank
    run's, both on K15 (kernels_uniwin.h, packed2 mode 8) - the kernel, the slab offsets and the halo arithmetic the 8-GPU run uses."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    if _TORCH_COLD:
        pytest.skip("torch did not come up in a child process on this box (see test_device_buffer_entry_point_with_torch)")
    root = Path(__file__).resolve().parent.parent
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    common = ["--workload", "c3", "--c3-n", "16", "--steps", "3", "--warmup", "1", "--no-pmc"]
    try:
        one = subprocess.run([sys.executable, str(root / "bench.py")] + common, capture_output=True, text=True, timeout=600)
        assert one.returncode == 0, one.stderr[-2000:]
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), str(root / "bench.py"),
                              "--gpus", "2", "--backend", "gloo", "--share-gpu"] + common,
                             capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        _TORCH_COLD.append(1)
        pytest.skip("a torch child process did not finish within 600 s on this box (cold image)")
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    b = json.loads(two.stdout.strip().splitlines()[-1])
    assert a["config"]["states"] == b["config"]["states"] == 16 ** 6 and a["config"]["controls"] == 1331
    assert b["n_gpus"] == 2 and "8 of 16 planes per GPU, halo 0/1" in b["config"]["sharding"]
    assert a["config"]["kernel_variant"] == 4 and b["config"]["kernel_variant"] == 4
    assert "cpu_baseline" not in a and "other_workloads" not in a          # the 176 GB configuration as the headline: nothing rides on it
    assert abs(a["checksum_sum_J"] - b["checksum_sum_J"]) <= 1e-12 * abs(a["checksum_sum_J"]) and a["checksum_sum_J"] > 0


F16 = [((9, 8), (3,), False), ((13, 11, 9), (4, 5), True), ((6, 5, 4, 5), (3, 4), False), ((7, 6), (70,), True)]


@pytest.mark.parametrize("n,m,nonuniform", F16)
def test_float16_j_storage_bit_exact(env, n, m, nonuniform):
    """HJB_F16S (BASELINE config 'fp16 cost-to-go storage'): J buffers are IEEE half, arithmetic is
    float32, the store rounds to nearest even.  Bit-exact against the oracle (F16C conversions) for
    every kernel that supports it (0, 4, 5); halves really are half the bytes."""
    hjbdp, _abi, c_oracle = env
    from problems import nested_problem, random_terminal
    s32 = nested_problem(31 + len(n), n, m, dtype=np.float32, nonuniform=nonuniform, spread=0.3)
    s16 = hjbdp.ProblemSpec(s32.knots, s32.m, s32.next_terms, s32.cost_terms, dtype=np.float32, index_base=1,
                            j_storage=np.float16)
    term = random_terminal(s32, 2).astype(np.float16)
    ref = c_oracle.sweep(_abi, s16, 4, terminal=term, keep_J=True, keep_idx=True)
    assert ref["J"].dtype == np.float16
    seen = set()
    for v in (None, 0, 4, 5):
        try:
            bk = hjbdp.Backup(s16, variant=v)
        except hjbdp.HjbError as e:
            assert v == 4 and e.status == _abi.HJB_E_UNSUPPORTED
            continue
        with bk:
            seen.add(bk.info()["kernel_variant"])
            o = bk.solve(4, terminal=term, keep_J=True, keep_idx=True)
        assert o["J"].dtype == np.float16
        assert np.array_equal(o["J_stages"].view(np.uint16), ref["J_stages"].view(np.uint16)), v
        assert np.array_equal(o["idx_stages"], ref["idx_stages"]), v
    assert seen <= {0, 4, 5} and 0 in seen
    with pytest.raises(hjbdp.HjbError) as ei:     # unsupported storage for this kernel: refused, loudly
        hjbdp.Backup(s16, variant=1)
    assert ei.value.status == _abi.HJB_E_UNSUPPORTED
    # float16 storage tracks float32 storage to half precision
    with hjbdp.Backup(s32) as bk:
        o32 = bk.solve(4, terminal=term.astype(np.float32))
    err = np.abs(o["J"].astype(np.float32) - o32["J"])
    assert np.median(err) < 2e-3 * np.abs(o32["J"]).max() and err.max() < 5e-2 * np.abs(o32["J"]).max()


def test_handles_on_different_threads_overlap_safely(env):
    """include/hjbdp.h threading contract: different handles may be driven from different host threads at once
    (graph capture, monitor read-backs, allocation and synchronous copies all in flight).  Six sweeps with
    graph replay and the early-stop monitor, twice; every result equals the oracle's."""
    hjbdp, _abi, c_oracle = env
    from problems import nested_problem, random_terminal
    specs = [nested_problem(500 + i, (9 + i, 8, 6), (3, 4), dtype=np.float32, spread=0.1) for i in range(6)]
    terms = [random_terminal(sp, i) for i, sp in enumerate(specs)]
    refs = [c_oracle.sweep(_abi, sp, 150, terminal=t, monitor_period=10, monitor_tol=1e-9, nthreads=4)
            for sp, t in zip(specs, terms)]        # tiny problems: a 128-thread OpenMP team costs more than the work
    for _ in range(2):
        outs, wall_ms, variants = hjbdp.solve_many(specs, 150, terminal=terms, monitor_period=10, monitor_tol=1e-9)
        for o, r in zip(outs, refs):
            assert o["stages_done"] == r["stages_done"]
            assert np.array_equal(o["J"], r["J"]) and np.array_equal(o["idx"], r["idx"])


@pytest.mark.parametrize("dtype,j_storage,n,stages", [
    (np.float64, None, (41, 37), 37), (np.float32, None, (70, 20), 100), (np.float32, np.float16, (33, 50), 29),
    (np.float64, None, (16, 16), 16), (np.float64, None, (5, 90), 70)])
def test_temporal_blocking_2d_bit_exact(env, dtype, j_storage, n, stages):
    """K9 (kernels_tile2d.h): several stages per launch with the J patch resident in LDS must leave exactly what
    one launch per stage leaves - ragged tiles, stage counts that are not multiples of the block, graph replay,
    float16 storage rounding at every stage."""
    hjbdp, _abi, c_oracle = env
    from problems import Term
    rng = np.random.default_rng(n[0])
    kx, kv = np.linspace(-0.5, 0.5, n[0]), np.linspace(-0.4, 0.6, n[1])
    hx, hv = kx[1] - kx[0], kv[1] - kv[0]
    U = np.array([-0.26, 0.0, 0.13, 0.26])
    # x+ = x + a(v) with |a| < one cell, v+ = v + b(x) + c(u) with |b| + |c| < one cell: local dynamics
    nxt = [[Term((0,), kx), Term((1,), 0.9 * hx * np.sin(3 * kv))],
           [Term((1,), kv), Term((0,), 0.4 * hv * np.cos(5 * kx)), Term((2,), 0.55 * hv * U / 0.26)]]
    cost = [Term((0,), 6 * kx ** 2), Term((1,), 3 * kv ** 2), Term((2,), 0.1 * U ** 2), Term((0, 1), 0.05 * rng.random(n))]
    spec = hjbdp.ProblemSpec([kx, kv], [len(U)], nxt, cost, dtype=dtype, index_base=1, j_storage=j_storage)
    term = (rng.random(spec.nS) * 2).astype(spec.j_dtype)
    ref = c_oracle.sweep(_abi, spec, stages, terminal=term, nthreads=8)
    with hjbdp.Backup(spec) as bk:
        bk.set_option("temporal", 2)                 # fail if the blocked path is not the one that runs
        out = bk.solve(stages, terminal=term)
        assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
        bk.set_option("temporal", 0)
        out0 = bk.solve(stages, terminal=term)
        assert np.array_equal(out0["J"], ref["J"]) and np.array_equal(out0["idx"], ref["idx"])


@pytest.mark.parametrize("dtype,j_storage,n,m,stages", [
    (np.float64, None, (41, 37), (9,), 37),                 # 9 controls: more than the cached form keeps in registers
    (np.float32, None, (70, 20), (3, 3), 100),              # two control dims (labels enumerate them column-major); graph replay
    (np.float32, np.float16, (33, 50), (5, 2), 29),         # float16 storage, ragged tiles, stages % 8 != 0
    (np.float64, None, (16, 16), (2, 2, 3), 16),            # three control dims
    (np.float32, None, (5, 90), (17,), 70)])
def test_temporal_blocking_2d_uncached_form_bit_exact(env, dtype, j_storage, n, m, stages):
    """K9's general form `k_backup_tile2d` (more than 4 controls or more than one control dim: per-control table
    look-ups inside the multi-stage kernel, label composition over the control dims) - the default for such local 2-D
    problems, so it needs the same coverage as the cached form above."""
    hjbdp, _abi, c_oracle = env
    from problems import Term
    rng = np.random.default_rng(n[0] + len(m))
    kx, kv = np.linspace(-0.5, 0.5, n[0]), np.linspace(-0.4, 0.6, n[1])
    hx, hv = kx[1] - kx[0], kv[1] - kv[0]
    C = len(m)
    # x+ = x + a(v) + a small control push, v+ = v + b(x) + sum of per-control-dim pushes: every query within one cell
    us = [np.linspace(-1.0, 1.0, mc) if mc > 1 else np.zeros(1) for mc in m]
    nxt = [[Term((0,), kx), Term((1,), 0.6 * hx * np.sin(3 * kv)), Term((2,), 0.3 * hx * us[0])],
           [Term((1,), kv), Term((0,), 0.3 * hv * np.cos(5 * kx))] + [Term((2 + c,), (0.6 / C) * hv * us[c]) for c in range(C)]]
    cost = [Term((0,), 6 * kx ** 2), Term((1,), 3 * kv ** 2)] + [Term((2 + c,), 0.1 * (1 + c) * np.round(us[c] * 2) ** 2) for c in range(C)]
    cost.append(Term((0, 2), 0.05 * rng.random((n[0], m[0]))))
    spec = hjbdp.ProblemSpec([kx, kv], list(m), nxt, cost, dtype=dtype, index_base=1, j_storage=j_storage)
    term = (rng.random(spec.nS) * 2).astype(spec.j_dtype)
    ref = c_oracle.sweep(_abi, spec, stages, terminal=term, nthreads=8)
    with hjbdp.Backup(spec) as bk:
        bk.set_option("temporal", 2)                 # fail if the blocked path is not the one that runs
        out = bk.solve(stages, terminal=term)
        assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
        assert out["idx"].min() >= 1 and out["idx"].max() <= spec.nU


def test_temporal_blocking_refused_when_not_local(env):
    """Kirk's dynamics move x2 by up to ~40 cells per stage: K9 must not be used (and says so when required)."""
    hjbdp, _abi, c_oracle = env
    spec = _kirk(hjbdp, "double", 20, 25, 30).build_spec()
    with hjbdp.Backup(spec) as bk:
        out = bk.solve(20)                            # default: falls back silently to one launch per stage
        ref = c_oracle.sweep(_abi, spec, 20, nthreads=8)
        assert np.array_equal(out["J"], ref["J"])
        bk.set_option("temporal", 2)
        with pytest.raises(hjbdp.HjbError) as ei:
            bk.solve(20)
        assert ei.value.status == _abi.HJB_E_UNSUPPORTED


@pytest.mark.extended        # (12 s of random shapes: opt-in; tools/stress_parity.py runs the long form, profiles/r0N_stress_parity.txt)
@pytest.mark.order(90)
@pytest.mark.watchdog(400)
def test_randomised_stress_slice(env):
    """12 seconds of tools/stress_parity.py (random shapes, every applicable stage-kernel variant, multi-stage paths,
    slabs): everything that stays finite must equal the oracle bit for bit."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_parity.py"), "12", "5"],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "stress ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("case", ["c2_small", "pos_att", "kirk", "nested6", "colsweep_slab"])
def test_mfma_table_build_is_bit_identical(env, case):
    """BASELINE config 5's "batched-GEMM A.X path on MFMA": the only place the affine next-state sum over all grid
    states exists in this design is the one-off table build; its MFMA form (v_mfma_f32_32x32x2_f32 as the outer sum
    [f 1].[1; g], csrc/kernels_prep_mfma.h) must give the SAME tables as the vector build - the f32 MFMA is a k-ordered
    fmaf chain, so fma(1, g, fma(f, 1, 0)) = round(f + g): equal bits, checked on the tables' hash and on a sweep."""
    hjbdp, _abi, c_oracle = env
    from problems import colsweep_problem, nested_problem, random_terminal
    from hjbdp.synthetic import position3d_spec
    slab, variant = None, None
    if case == "c2_small":
        spec = position3d_spec(n=23, mu=7)                                    # variant 4: axis tables built in hjb_create
    elif case == "pos_att":
        pa = hjbdp.Solver_pos_att()
        pa.cost_mode = "terms"
        pa.table_dtype = None                # float32 queries: the float64 build has no MFMA form
        pa.n_mesh_x, pa.n_mesh_v, pa.n_mesh_t, pa.n_mesh_w = 37, 9, 8, 11
        sx, sv, st, sw = pa.grids()
        spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7,
                                        pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
        variant = 5
    elif case == "kirk":
        spec, variant = _kirk(hjbdp, "single", 8, 45, 70).build_spec(), 5     # rows (x1, x2), columns u: 70 > one 32-wide tile
    elif case == "nested6":
        spec, variant = nested_problem(77, (4, 3, 4, 3, 3, 5), (3, 3, 3), dtype=np.float32), 5
    else:
        spec, variant, slab = colsweep_problem(9, (36, 7, 9, 14), gax=2), 7, (4, 9, 1, 1)
    term = random_terminal(spec, 3)
    with hjbdp.Backup(spec, slab=slab, variant=variant) as bk:
        if slab is None:
            ref = bk.solve(3, terminal=term)
        bk.set_option("prep_mfma", 0)
        h0, n_tab = bk.get_option("table_hash"), bk.get_option("prep_tables")
        bk.set_option("prep_mfma", 1)
        h1, n_mfma = bk.get_option("table_hash"), bk.get_option("prep_mfma_tables")
        assert n_tab >= 1 and n_mfma >= 1, (n_tab, n_mfma)       # the MFMA form applied to at least one axis
        assert h0 == h1
        if slab is None:
            out = bk.solve(3, terminal=term)
            assert np.array_equal(out["J"], ref["J"]) and np.array_equal(out["idx"], ref["idx"])
            orc = c_oracle.sweep(_abi, spec, 3, terminal=term)
            assert np.array_equal(out["J"], orc["J"]) and np.array_equal(out["idx"], orc["idx"])


def test_runaway_sweep_nan_totals_follow_the_sequential_rule(env):
    """A case tools/stress_parity.py found (seed 11): a random 60x9x9x8 x 10 float32 problem whose cost-to-go runs away to
    -2.7e38 over 70 stages; states pass through -inf, totals become NaN (inf - inf).  Outside the contract, but the
    control-split kernel must not let a NaN held by one lane hide that lane's subtree of the reduction: like the
    sequential rule `first || tot < best` it skips NaN candidates (and keeps a NaN only when control 0's total is one)."""
    hjbdp, _abi, c_oracle = env
    import pickle
    from hjbdp import Term
    with open(ROOT_GOLDEN / "runaway_60x9x9x8.pkl", "rb") as fh:
        d = pickle.load(fh)
    spec = hjbdp.ProblemSpec(d["knots"], d["m"], [[Term(dims, data) for dims, data in ts] for ts in d["next_terms"]],
                             [Term(dims, data) for dims, data in d["cost_terms"]], dtype=np.float32, index_base=1, idx_dtype=np.uint16)
    ref = c_oracle.sweep(_abi, spec, d["stages"], terminal=d["term"], nthreads=8)
    for v in (0, 3, 5):
        with hjbdp.Backup(spec, variant=v) as bk:
            out = bk.solve(d["stages"], terminal=d["term"])
        assert np.array_equal(out["J"], ref["J"], equal_nan=True) and np.array_equal(out["idx"], ref["idx"]), v
