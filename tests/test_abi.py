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
    expect += 4 + 4 + 8 + 8 * 4             # model, reserved1, model_h, model_tables[4]
    assert C.sizeof(_abi.hjb_problem) == expect
    assert C.sizeof(_abi.hjb_solve_opts) == 4 + 4 + 8 + 8 * 5 + 8 + 8 + 8 + 4 + 4
    assert C.sizeof(_abi.hjb_probe) == 4 * 6 * 2 + 4 * 3 + 4 + 8 * 3
    assert C.sizeof(_abi.hjb_result) == 32
    assert C.sizeof(_abi.hjb_info) == 48


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
