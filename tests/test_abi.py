"""CPU tests of the drop-in boundary: the shared library loads without a GPU,
exports every symbol include/hjbdp.h declares, and argument errors come back as
status codes (never exceptions across the ABI)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib(built):
    import hjbdp
    return hjbdp.load_library()


def test_every_declared_symbol_is_exported(lib):
    from hjbdp import _abi
    header = (ROOT / "include" / "hjbdp.h").read_text()
    declared = set(re.findall(r"\b(hjb_[a-z_0-9]+)\s*\(", header)) - {"hjb_progress_fn"}
    assert declared == set(_abi.SYMBOLS), declared ^ set(_abi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_status_strings(lib):
    assert b"hjbdp" in lib.hjb_version()
    assert lib.hjb_status_string(0) == b"ok"
    assert lib.hjb_status_string(5) == b"query outside slab halo"


def test_struct_layout_matches_header(lib):
    """sizes derived by hand from include/hjbdp.h (LP64)."""
    from hjbdp import _abi
    assert C.sizeof(_abi.hjb_term) == 16
    expect = 4 * 2 + 4 * 6 + 4 * 3 + 4 * 2  # D,C,n,m,dtype,index_base = 52 -> pad to 56
    expect = 56 + 8 * 6 + 4 * 6 + 16 * 12 * 6 + 8 + 16 * 12 + 16
    expect += 4 + 4 + 8 + 8 * 4             # model, table_dtype, model_h, model_tables[4]
    expect += 4 + 4                         # cost_dtype, reserved_
    assert C.sizeof(_abi.hjb_problem) == expect
    assert C.sizeof(_abi.hjb_solve_opts) == 4 + 4 + 8 + 8 * 5 + 8 + 8 + 8 + 4 + 4
    assert C.sizeof(_abi.hjb_probe) == 4 * 6 * 2 + 4 * 3 + 4 + 8 * 3
    assert C.sizeof(_abi.hjb_result) == 32
    assert C.sizeof(_abi.hjb_info) == 64


def test_invalid_problems_are_rejected_with_status(lib):
    import hjbdp
    from hjbdp import _abi
    k = np.linspace(0, 1, 5)
    spec = hjbdp.ProblemSpec([k], [3], [[hjbdp.Term((0,), k)]], [hjbdp.Term((0,), k)])
    p, keep = spec.to_c()
    h = C.c_void_p()
    p.D = 9
    assert lib.hjb_create(C.byref(p), 0, C.byref(h)) == _abi.HJB_E_UNSUPPORTED
    assert b"D=9" in lib.hjb_last_error(None)
    p, keep = spec.to_c()
    bad = np.array([0.0, 0.5, 0.5, 0.75, 1.0])
    p.knots[0] = bad.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.hjb_create(C.byref(p), 0, C.byref(h)) == _abi.HJB_E_INVALID
    p, keep = spec.to_c()
    p.index_base = 2
    assert lib.hjb_create(C.byref(p), 0, C.byref(h)) == _abi.HJB_E_INVALID
    p, keep = spec.to_c((3, 2, 0, 0))
    assert lib.hjb_create(C.byref(p), 0, C.byref(h)) == _abi.HJB_E_INVALID
    assert lib.hjb_create(None, 0, C.byref(h)) == _abi.HJB_E_INVALID
    assert lib.hjb_destroy(None) == _abi.HJB_OK


def test_flat_builder_validates_without_a_gpu(lib):
    """The flat builder API (what a MATLAB host binds with calllib: primitives and plain arrays only) checks its
    arguments on the host; only hjb_create_from needs a device."""
    from hjbdp import _abi
    b = C.c_void_p()
    n = (C.c_int32 * 2)(5, 4)
    m = (C.c_int32 * 1)(3)
    assert lib.hjb_problem_new(2, 1, n, m, _abi.HJB_F64, 1, C.byref(b)) == _abi.HJB_OK
    k = np.linspace(0, 1, 5)
    assert lib.hjb_problem_set_knots(b, 0, k.ctypes.data_as(C.POINTER(C.c_double)), 5) == _abi.HJB_OK
    assert lib.hjb_problem_set_knots(b, 1, k.ctypes.data_as(C.POINTER(C.c_double)), 5) == _abi.HJB_E_INVALID   # 4 points
    assert b"4 grid points" in lib.hjb_problem_last_error(b)
    assert lib.hjb_problem_set_knots(b, 2, k.ctypes.data_as(C.POINTER(C.c_double)), 5) == _abi.HJB_E_INVALID
    t = np.zeros(20)
    assert lib.hjb_problem_add_next_term(b, 0, 0b011, t.ctypes.data, 20) == _abi.HJB_OK
    assert lib.hjb_problem_add_next_term(b, 0, 0b011, t.ctypes.data, 19) == _abi.HJB_E_INVALID   # wrong element count
    assert lib.hjb_problem_add_next_term(b, 0, 0b1000, t.ctypes.data, 1) == _abi.HJB_E_INVALID   # dim 3 does not exist
    assert lib.hjb_problem_add_cost_term(b, 0b100, t.ctypes.data, 3) == _abi.HJB_OK
    h = C.c_void_p()
    st = lib.hjb_create_from(b, 0, C.byref(h))
    assert st == _abi.HJB_E_INVALID and b"knots of axis 1" in lib.hjb_problem_last_error(b)
    assert lib.hjb_problem_free(b) == _abi.HJB_OK
    assert lib.hjb_problem_new(7, 1, n, m, _abi.HJB_F64, 1, C.byref(b)) == _abi.HJB_E_UNSUPPORTED
    assert lib.hjb_problem_free(None) == _abi.HJB_OK


def _builder_from_spec(lib, spec):
    """A ProblemSpec through the flat builder calls (float32 / float64 arithmetic, no model)."""
    from hjbdp import _abi
    b = C.c_void_p()
    n = (C.c_int32 * spec.D)(*spec.n)
    m = (C.c_int32 * spec.C)(*spec.m)
    dt = _abi.HJB_F64 if spec.dtype == np.float64 else _abi.HJB_F32
    assert lib.hjb_problem_new(spec.D, spec.C, n, m, dt, spec.index_base, C.byref(b)) == _abi.HJB_OK
    for a in range(spec.D):
        k = np.ascontiguousarray(spec.knots[a], dtype=np.float64)
        assert lib.hjb_problem_set_knots(b, a, k.ctypes.data_as(C.POINTER(C.c_double)), k.size) == _abi.HJB_OK
    idx_code = {None: _abi.HJB_IDX_I32, "auto": _abi.HJB_IDX_AUTO}.get(spec.idx_dtype if not hasattr(spec.idx_dtype, "itemsize") else None,
                                                                    {1: _abi.HJB_IDX_U8, 2: _abi.HJB_IDX_U16, 4: _abi.HJB_IDX_I32}[spec.idx_np_dtype.itemsize])
    if spec.table_dtype is not None or spec.idx_dtype is not None:
        assert lib.hjb_problem_set_types(b, idx_code, _abi.HJB_TAB_F64 if spec.table_dtype is not None else _abi.HJB_TAB_DEFAULT) == _abi.HJB_OK

    def flat(t, dtype=None):
        mask = sum(1 << d for d in t.dims)
        v = np.ascontiguousarray(np.asarray(t.data, dtype=dtype or spec.dtype).reshape(-1, order="F"))
        return mask, v
    for a in range(spec.D):
        for t in spec.next_terms[a]:
            mask, v = flat(t, spec.table_dtype)
            assert lib.hjb_problem_add_next_term(b, a, mask, v.ctypes.data, v.size) == _abi.HJB_OK
    for t in spec.cost_terms:
        mask, v = flat(t)
        assert lib.hjb_problem_add_cost_term(b, mask, v.ctypes.data, v.size) == _abi.HJB_OK
    return b


def test_flat_builder_suggests_the_fast_axis_order_for_pos_att(lib):
    """hjb_problem_suggest_order on the reference's own Solver_pos_att channel (axes x, v, theta, w as
    Solver_pos_att.m:299-328 builds them): the labelling of the column-sweep kernel, (x, theta, w, v) = old axes
    (0, 2, 3, 1) - what hjbdp.Solver_pos_att.FAST_AXIS_ORDER + the bench use; nothing to suggest for a problem that is
    already labelled that way, for Kirk's 2-D problem, or when every axis moves with the control.  No GPU involved."""
    import hjbdp
    from hjbdp import _abi
    pa = hjbdp.Solver_pos_att()
    pa.cost_mode = "terms"
    sx, sv, st, sw = pa.grids()
    spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1,
                                    pa.Qw1, pa.R1, pa.J2)
    b = _builder_from_spec(lib, spec)
    order = (C.c_int32 * 4)()
    found = C.c_int32(-1)
    assert lib.hjb_problem_suggest_order(b, order, C.byref(found)) == _abi.HJB_OK
    assert found.value == 1 and tuple(order) == (0, 2, 3, 1)
    assert lib.hjb_problem_permute_axes(b, order) == _abi.HJB_OK
    assert lib.hjb_problem_suggest_order(b, order, C.byref(found)) == _abi.HJB_OK
    assert found.value == 0 and tuple(order) == (0, 1, 2, 3)                  # already labelled that way
    bad = (C.c_int32 * 4)(0, 1, 1, 3)
    assert lib.hjb_problem_permute_axes(b, bad) == _abi.HJB_E_INVALID and b"permutation" in lib.hjb_problem_last_error(b)
    assert lib.hjb_problem_set_slab(b, 2, 5, 1, 1) == _abi.HJB_OK
    assert lib.hjb_problem_permute_axes(b, (C.c_int32 * 4)(0, 1, 2, 3)) == _abi.HJB_E_INVALID       # after a slab: refused
    assert lib.hjb_problem_free(b) == _abi.HJB_OK
    # Solver_attitude.run's 6-D problem in the reference's dim order (w1, w2, w3, yaw, pitch, roll): angles first, the
    # rates in the order of their torques - what the mirror's AXIS_ORDER applies
    sa = hjbdp.Solver_attitude()
    sa.n_mesh_w, sa.n_mesh_q = 4, 5
    b3 = _builder_from_spec(lib, sa.build_spec_full())
    o6 = (C.c_int32 * 6)()
    assert lib.hjb_problem_suggest_order(b3, o6, C.byref(found)) == _abi.HJB_OK
    assert found.value == 1 and tuple(o6) == tuple(hjbdp.Solver_attitude.AXIS_ORDER) == (3, 4, 5, 0, 1, 2)
    assert lib.hjb_problem_free(b3) == _abi.HJB_OK
    from problems import random_problem
    b2 = _builder_from_spec(lib, random_problem(3, (5, 4, 3, 4), (3,), dtype=np.float32))   # every axis sees the control
    assert lib.hjb_problem_suggest_order(b2, order, C.byref(found)) == _abi.HJB_OK and found.value == 0
    assert lib.hjb_problem_free(b2) == _abi.HJB_OK


def _prototypes(text):
    """name -> normalised parameter-type list of every hjb_* prototype in a header."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(hjb_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = []
        for a in m.group(2).split(","):
            a = re.sub(r"\b(const|struct)\b", " ", a)
            a = re.sub(r"\b(hjb_builder|hjb_handle|hjb_multi|hjb_rank)\b", "void *", a)      # opaque handles
            a = re.sub(r"[A-Za-z_][A-Za-z_0-9]*\s*$", "", a.strip()) if not a.strip().endswith("*") and a.strip() != "void" else a
            args.append(re.sub(r"\s+", "", a))
        out[m.group(1)] = args
    return out


def test_matlab_header_is_the_flat_subset_of_the_c_header(lib):
    """include/hjbdp_matlab.h (what loadlibrary parses: no structs, handles as void *) must declare a subset of
    include/hjbdp.h's entry points with the same parameter lists, all exported; and nothing that takes a struct."""
    full = _prototypes((ROOT / "include" / "hjbdp.h").read_text())
    flat_text = (ROOT / "include" / "hjbdp_matlab.h").read_text()
    flat = _prototypes(flat_text)
    assert len(flat) >= 20 and "struct" not in re.sub(r"/\*.*?\*/", " ", flat_text, flags=re.S)
    for name, args in flat.items():
        assert hasattr(lib, name), name
        assert name in full, name
        assert args == full[name], (name, args, full[name])
    for name in ("hjb_problem_new", "hjb_create_from", "hjb_solve_flat", "hjb_create_multi_from", "hjb_solve_multi_flat"):
        assert name in flat
    for name in ("hjb_rank_create_from", "hjb_rank_stage", "hjb_rank_info", "hjb_device_malloc", "hjb_device_copy"):
        assert name in flat                          # a MATLAB worker per GPU can bind the rank API
    # every calllib in every .m file: a function the flat header declares, called with as many arguments as it takes
    mdir = ROOT / "optimal-control-dynamic-programming_amd" / "matlab"
    n_calls = 0
    for mfile in sorted(mdir.glob("*.m")):
        for name, nargs in _matlab_calllibs(mfile.read_text()):
            assert name in flat, (mfile.name, name)
            want = 0 if flat[name] == ["void"] else len(flat[name])
            assert nargs == want, (mfile.name, name, nargs, flat[name])
            n_calls += 1
    assert n_calls >= 18


def _matlab_calllibs(text):
    """(function name, number of arguments after the name) of every calllib(L, 'hjb_...', ...) in MATLAB source."""
    text = re.sub(r"\.\.\.[^\n]*\n", " ", text)            # line continuations
    code = "\n".join(l if not l.lstrip().startswith("%") else "" for l in text.split("\n"))
    out = []
    for m in re.finditer(r"calllib\(L,\s*'(hjb_[a-z_0-9]+)'", code):
        i, depth, nargs, in_str = m.end(), 1, 0, False
        while depth > 0:
            c = code[i]
            if in_str:
                in_str = c != "'"
            elif c == "'" and not (code[i - 1].isalnum() or code[i - 1] in ")]}_."):     # a quote that opens a string, not a transpose
                in_str = True
            elif c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                nargs += 1
            elif c == "\n":
                raise AssertionError("unterminated calllib: " + code[m.start():m.start() + 80])
            i += 1
        out.append((m.group(1), nargs))
    return out


def test_ctypes_twin_calls_what_the_matlab_shim_calls():
    """tests/matlab_twin.py::hjbdp_solve is the stand-in the GPU tests run for matlab/hjbdp_solve.m: both must drive the same
    set of flat entry points (names; the arities of the .m side are checked above, ctypes checks its own)."""
    shim = (ROOT / "optimal-control-dynamic-programming_amd" / "matlab" / "hjbdp_solve.m").read_text()
    twin = (ROOT / "tests" / "matlab_twin.py").read_text()
    twin = twin[twin.index("def hjbdp_solve("):twin.index("# the reference classes' helper methods")]
    in_m = {name for name, _ in _matlab_calllibs(shim)}
    in_py = set(re.findall(r"lib\.(hjb_[a-z_0-9]+)", twin))
    assert in_m == in_py, in_m ^ in_py


def test_matlab_solver_shims_cover_the_reference_methods():
    """north_star: 'Host code stays in MATLAB'.  One .m body per reference method on the path, each citing the lines it
    replaces, each ending in the properties / files the reference method leaves, each on hjbdp_solve (the one file that
    touches calllib)."""
    mdir = ROOT / "optimal-control-dynamic-programming_amd" / "matlab"
    want = {
        "Dynamic_Solver_hjbdp_run.m": ("test/Dynamic_Solver.m:66-105", ["obj.u_star", "obj.J_star", "obj.F = griddedInterpolant"]),
        "Solver_position_hjbdp_simplified_run.m": ("position-control/Solver_position.m:94-150",
                                                   ["obj.U1_Opt = pol", "obj.U3_Opt = pol", "'nearest'", "obj.n_mesh_x = length"]),
        "Solver_attitude_hjbdp_simplified_run.m": ("attitude-control/Solver_attitude.m:196-259",
                                                   ["obj.U1_Opt = pol", "obj.U3_Opt = pol", "'nearest'"]),
        "Solver_attitude_hjbdp_run.m": ("attitude-control/Solver_attitude.m:261-300",
                                        ["obj.F = griddedInterpolant", "obj.U1_Opt = single(obj.U_vector", "obj.U3_Opt = single(obj.U_vector",
                                         "prob.model"]),
        "Solver_pos_att_hjbdp_channel.m": ("pos-att/Solver_pos_att.m:244-297",
                                           ["save(file_name, 'F_gI', 'U_Optimal_id', 'f0_allcomb', 'f1_allcomb', 'f6_allcomb', 'f7_allcomb')",
                                            "'double_tables', true", "'monitor_single', true", "'monitor_period', 50"]),
        "Solver_pos_att_hjbdp_simplified_run.m": ("pos-att/Solver_pos_att.m:197-242", ["channel_x_controller_1_failure"]),
    }
    for name, (cite, needles) in want.items():
        text = (mdir / name).read_text()
        assert cite in text, (name, cite)
        for nd in needles:
            assert nd in text, (name, nd)
        if name != "Solver_pos_att_hjbdp_simplified_run.m":
            assert "hjbdp_solve(" in text and "calllib" not in text.replace("calllib can", ""), name
    # the reference files the shims cite exist where they say (this container only: the GPU box has no /root/reference)
    ref = Path("/root/reference")
    if ref.exists():
        for rel in ("test/Dynamic_Solver.m", "position-control/Solver_position.m", "attitude-control/Solver_attitude.m",
                    "pos-att/Solver_pos_att.m"):
            assert (ref / rel).exists(), rel


def test_mex_gateway_compiles_against_the_c_header():
    """mex/hjbdp_mex.c cannot be built into a MEX file here (no MATLAB, no mex.h); it is syntax- and type-checked
    against include/hjbdp.h with a test-only declaration stub of the MEX API (tests/mex_stub/mex.h)."""
    import subprocess
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "tests" / "mex_stub"),
                        "-I", str(ROOT / "include"), str(ROOT / "mex" / "hjbdp_mex.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = (ROOT / "mex" / "hjbdp_mex.c").read_text()
    assert "mexFunction" in src and "hjb_create_from" in src and "hjb_solve_flat" in src


def test_no_gpu_means_loud_failure_not_fallback(lib):
    """Without a HIP device hjb_create must fail with HJB_E_DEVICE."""
    import hjbdp
    from hjbdp import _abi
    if hjbdp.device_count() > 0:
        pytest.skip("a GPU is visible")
    k = np.linspace(0, 1, 5)
    spec = hjbdp.ProblemSpec([k], [3], [[hjbdp.Term((0,), k)]], [hjbdp.Term((0,), k)])
    with pytest.raises(hjbdp.HjbError) as ei:
        hjbdp.Backup(spec)
    assert ei.value.status == _abi.HJB_E_DEVICE
    ds = hjbdp.Dynamic_Solver()
    ds.N, ds.dx, ds.du = 3, 4, 5
    with pytest.raises(hjbdp.HjbError):
        ds.run()


def test_solve_batch_checks_its_arguments_before_any_device_work(lib):
    """hjb_solve_batch (include/hjbdp.h): a null or empty batch is HJB_E_INVALID, more than eight problems HJB_E_UNSUPPORTED - decided
    on the host, before a device is touched (no GPU here); the Python entry point without a device fails loudly like Backup does."""
    import ctypes as C
    import hjbdp
    from hjbdp import _abi
    lib.hjb_solve_batch.restype = C.c_int32
    assert lib.hjb_solve_batch(0, None, None, None) == _abi.HJB_E_INVALID
    assert lib.hjb_solve_batch(2, None, None, None) == _abi.HJB_E_INVALID
    hs = (C.c_void_p * 9)()
    op = (C.POINTER(_abi.hjb_solve_opts) * 9)()
    assert lib.hjb_solve_batch(9, hs, op, None) == _abi.HJB_E_UNSUPPORTED
    assert lib.hjb_solve_batch(2, hs, op, None) == _abi.HJB_E_INVALID          # null handles
    if hjbdp.device_count() == 0:
        k = np.linspace(0, 1, 5)
        spec = hjbdp.ProblemSpec([k], [3], [[hjbdp.Term((0,), k)]], [hjbdp.Term((0,), k)])
        with pytest.raises(hjbdp.HjbError) as ei:
            hjbdp.solve_batch([spec, spec], 3)
        assert ei.value.status == _abi.HJB_E_DEVICE


def test_host_without_librccl_gets_a_status_not_a_crash(built):
    """with no librccl to dlopen (here: the loader restricted to $HJBDP_RCCL_LIB, which names nothing),
    hjb_rank_comm_unique_id returns HJB_E_UNSUPPORTED and the loader's message - it used to build that message from a
    second dlerror() call (NULL) and crash.  In a child process: the loader caches a library once it has one.  No GPU needed."""
    import subprocess
    import sys
    code = (
        "import sys, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "import hjbdp\n"
        "from hjbdp import _abi\n"
        "lib = hjbdp.load_library()\n"
        "assert lib.hjb_test_hook(b'rccl_only_env', 1) == _abi.HJB_OK\n"
        "uid = (C.c_char * 128)()\n"
        "st = lib.hjb_rank_comm_unique_id(uid)\n"
        "msg = (lib.hjb_rank_last_error(None) or b'').decode()\n"
        "print(st, msg)\n"
        "assert st == _abi.HJB_E_UNSUPPORTED, st\n"
        "assert 'dlopen' in msg and 'no/such' in msg, msg\n"
        "assert lib.hjb_rank_comm_unique_id(uid) == _abi.HJB_E_UNSUPPORTED\n"      # and again: still a status
        "assert lib.hjb_rank_comm_available() == _abi.HJB_E_UNSUPPORTED\n"        # the probe every rank asks before the set-up
        "assert 'dlopen' in (lib.hjb_rank_last_error(None) or b'').decode()\n"
    ) % str(ROOT / "optimal-control-dynamic-programming_amd")
    import os
    env = dict(os.environ, HJBDP_RCCL_LIB="/no/such/librccl.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may
    import, load or execute it (comments may mention it)."""
    pkg = ROOT / "optimal-control-dynamic-programming_amd"
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|libhjb_oracle|orc_backup_stage|orc_sweep|c_oracle|hjb_oracle\.py", re.M)
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")) + list(pkg.rglob("*.m")):
        assert not bad.search(f.read_text()), f


def test_problem_spec_validation():
    import hjbdp
    k = np.linspace(0, 1, 5)
    with pytest.raises(ValueError):
        hjbdp.Term((1, 0), np.zeros((2, 2)))
    with pytest.raises(ValueError):
        hjbdp.ProblemSpec([k], [3], [[hjbdp.Term((0,), np.zeros(4))]], [hjbdp.Term((0,), k)])
    with pytest.raises(ValueError):
        hjbdp.ProblemSpec([k], [3], [[hjbdp.Term((0,), k)]], [hjbdp.Term((0,), k)], dtype=np.float16)


def test_flat_builder_types_are_validated(lib):
    """hjb_problem_set_types: label storage and the table dtype of a problem under construction - the table dtype is the
    element type of the next-state terms, so it cannot change once one was added; float64 tables are for float32 problems."""
    from hjbdp import _abi
    n = (C.c_int32 * 2)(5, 4)
    m = (C.c_int32 * 1)(3)
    b = C.c_void_p()
    assert lib.hjb_problem_new(2, 1, n, m, _abi.HJB_F32, 1, C.byref(b)) == _abi.HJB_OK
    assert lib.hjb_problem_set_types(b, 7, 0) == _abi.HJB_E_INVALID and b"idx_dtype" in lib.hjb_problem_last_error(b)
    assert lib.hjb_problem_set_types(b, _abi.HJB_IDX_AUTO, 5) == _abi.HJB_E_INVALID
    assert lib.hjb_problem_set_types(b, _abi.HJB_IDX_U8, _abi.HJB_TAB_F64) == _abi.HJB_OK
    v = np.arange(5, dtype=np.float64)                    # float64 data now: 5 doubles
    assert lib.hjb_problem_add_next_term(b, 0, 1, v.ctypes.data, 5) == _abi.HJB_OK
    assert lib.hjb_problem_set_types(b, _abi.HJB_IDX_U8, _abi.HJB_TAB_DEFAULT) == _abi.HJB_E_INVALID      # terms already typed
    assert b"before adding" in lib.hjb_problem_last_error(b)
    assert lib.hjb_problem_set_types(b, _abi.HJB_IDX_U16, _abi.HJB_TAB_F64) == _abi.HJB_OK                # the label type may still change
    assert lib.hjb_problem_free(b) == _abi.HJB_OK
    b = C.c_void_p()
    assert lib.hjb_problem_new(2, 1, n, m, _abi.HJB_F64, 1, C.byref(b)) == _abi.HJB_OK
    assert lib.hjb_problem_set_types(b, _abi.HJB_IDX_I32, _abi.HJB_TAB_F64) == _abi.HJB_E_INVALID         # a float64 problem is float64 throughout
    assert lib.hjb_problem_free(b) == _abi.HJB_OK


def test_sizes_that_overflow_and_non_finite_data_are_refused_where_they_enter(lib):
    """six axes of 2^15 points wrapped the 64-bit state count to 0 and passed every size check; NaN / inf
    in knots, terms or model tables went through to the kernels, whose contract covers finite data only.  Both entry points
    (the struct API's hjb_create and the flat builder's setters) now answer with a status and name the place - without a GPU."""
    import hjbdp
    from hjbdp import _abi
    h = C.c_void_p()
    b = C.c_void_p()
    big = (C.c_int32 * 6)(*[1 << 15] * 6)
    m3 = (C.c_int32 * 3)(11, 11, 11)
    assert lib.hjb_problem_new(6, 3, big, m3, _abi.HJB_F32, 1, C.byref(b)) == _abi.HJB_E_UNSUPPORTED
    assert b"2^40" in lib.hjb_problem_last_error(None)
    k = np.linspace(0, 1, 5)
    spec = hjbdp.ProblemSpec([k, k], [3], [[hjbdp.Term((0,), k)], [hjbdp.Term((1,), k), hjbdp.Term((2,), np.arange(3.0))]],
                             [hjbdp.Term((0,), k), hjbdp.Term((2,), np.ones(3))])
    p, keep = spec.to_c()
    for a in range(6):
        p.n[a] = 1 << 15
    p.D = 6
    bigk = np.linspace(0, 1, 1 << 15)
    for a in range(6):
        p.knots[a] = bigk.ctypes.data_as(C.POINTER(C.c_double))
        p.n_next_terms[a] = 1
        p.next_terms[a][0] = p.next_terms[0][0]
        p.next_terms[a][0].mask = 0            # a scalar term: the sizes alone are on trial
    assert lib.hjb_create(C.byref(p), 0, C.byref(h)) == _abi.HJB_E_UNSUPPORTED
    assert b"2^40" in lib.hjb_last_error(None)
    # non-finite data, struct API
    for where, val, msg in (("knots", np.inf, b"knots[1][3] is not finite"), ("next", np.nan, b"next term 1 of axis 1: element 2 is not finite"),
                            ("cost", -np.inf, b"cost term 0: element 4 is not finite")):
        p, keep = spec.to_c()
        bad = {"knots": k.copy(), "next": np.arange(3.0), "cost": k.copy()}[where]
        if where == "knots":
            bad[3] = val
            p.knots[1] = bad.ctypes.data_as(C.POINTER(C.c_double))
        elif where == "next":
            bad[2] = val
            p.next_terms[1][1].data = bad.ctypes.data
        else:
            bad[4] = val
            p.cost_terms[0].data = bad.ctypes.data
        assert lib.hjb_create(C.byref(p), 0, C.byref(h)) == _abi.HJB_E_INVALID, where
        assert msg in lib.hjb_last_error(None), lib.hjb_last_error(None)
    # ... and the flat builder's setters
    n2 = (C.c_int32 * 2)(5, 5)
    m1 = (C.c_int32 * 1)(3)
    assert lib.hjb_problem_new(2, 1, n2, m1, _abi.HJB_F64, 1, C.byref(b)) == _abi.HJB_OK
    kb = k.copy(); kb[3] = np.inf
    assert lib.hjb_problem_set_knots(b, 0, kb.ctypes.data_as(C.POINTER(C.c_double)), 5) == _abi.HJB_E_INVALID
    assert b"knots of axis 0: element 3 is not finite" in lib.hjb_problem_last_error(b)
    kd = k.copy(); kd[2] = kd[1]
    assert lib.hjb_problem_set_knots(b, 0, kd.ctypes.data_as(C.POINTER(C.c_double)), 5) == _abi.HJB_E_INVALID
    assert b"not strictly increasing at 1" in lib.hjb_problem_last_error(b)
    t = np.arange(5.0); t[2] = np.nan
    assert lib.hjb_problem_add_next_term(b, 0, 0b001, t.ctypes.data, 5) == _abi.HJB_E_INVALID
    assert b"element 2 is not finite" in lib.hjb_problem_last_error(b)
    assert lib.hjb_problem_add_cost_term(b, 0b010, t.ctypes.data, 5) == _abi.HJB_E_INVALID
    assert lib.hjb_problem_free(b) == _abi.HJB_OK
    # the scan itself: float32 and float64, a hit in any block
    v = np.zeros(10000, dtype=np.float32); v[9999] = np.inf
    lib.hjb_problem_new(2, 1, (C.c_int32 * 2)(100, 100), m1, _abi.HJB_F32, 1, C.byref(b))
    assert lib.hjb_problem_add_next_term(b, 0, 0b011, v.ctypes.data, 10000) == _abi.HJB_E_INVALID
    assert b"element 9999" in lib.hjb_problem_last_error(b)
    v[9999] = 1.0
    assert lib.hjb_problem_add_next_term(b, 0, 0b011, v.ctypes.data, 10000) == _abi.HJB_OK
    assert lib.hjb_problem_free(b) == _abi.HJB_OK


def test_rccl_declarations_come_from_the_images_own_header():
    """csrc/hjbdp_rank.hip declared RCCL's ABI by hand (ncclUniqueId as 128 bytes by value, enum values, eight
    signatures).  It now includes <rccl/rccl.h> and takes every signature by decltype, so a drift fails the BUILD; this test holds
    the unit to that, and checks the facts the public header states about the id against the header the image ships."""
    src = (ROOT / "optimal-control-dynamic-programming_amd" / "csrc" / "hjbdp_rank.hip").read_text()
    assert "#include <rccl/rccl.h>" in src
    assert not re.search(r"kNccl\w+\s*=\s*\d", src), "an enum value written as a number"
    assert not re.search(r"\(\s*int\s*\(\s*\*\s*\)\s*\(", src), "a hand-written function-pointer cast"
    for f in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv",
              "ncclAllReduce", "ncclGetErrorString", "ncclCommCount", "ncclCommUserRank"):
        assert "decltype(&%s)" % f in src, f
        assert 'sym("%s")' % f in src or 'dlsym(lib, "%s")' % f in src, f
    assert "static_assert(NCCL_UNIQUE_ID_BYTES == 128" in src
    hdr = Path("/opt/rocm/include/rccl/rccl.h")
    if not hdr.exists():
        pytest.skip("no rccl.h in this image")
    h = hdr.read_text()
    assert re.search(r"#define\s+NCCL_UNIQUE_ID_BYTES\s+128\b", h)
    assert re.search(r"ncclCommInitRank\(ncclComm_t\*\s*comm,\s*int\s+nranks,\s*ncclUniqueId\s+commId,\s*int\s+rank\)", h)     # the id BY VALUE
    assert re.search(r"\bncclUint8\s*=\s*1\b", h) and re.search(r"\bncclFloat64\s*=\s*8\b", h) and re.search(r"\bncclSum\s*=\s*0\b", h)
    pub = (ROOT / "include" / "hjbdp.h").read_text()
    assert "128 bytes" in pub and "hjb_rank_comm_unique_id(void *id128_out)" in pub
