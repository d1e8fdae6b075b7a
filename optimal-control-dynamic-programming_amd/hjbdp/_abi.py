"""ctypes mirror of include/hjbdp.h (the C ABI of libhjbdp).

Field order and types must match the header exactly; tests/test_abi.py checks the
struct sizes against the compiled library's view and that every declared symbol
is exported.
"""
from __future__ import annotations

import ctypes as C

HJB_MAX_D = 6
HJB_MAX_C = 3
HJB_MAX_G = 9
HJB_MAX_TERMS = 12

HJB_OK = 0
HJB_E_INVALID = 1
HJB_E_UNSUPPORTED = 2
HJB_E_DEVICE = 3
HJB_E_NOMEM = 4
HJB_E_HALO = 5

HJB_F32 = 0
HJB_F64 = 1
HJB_F16S = 2

HJB_IDX_I32 = 0
HJB_IDX_U8 = 1
HJB_IDX_U16 = 2
HJB_IDX_AUTO = 3

HJB_TAB_DEFAULT = 0
HJB_TAB_F64 = 1

HJB_COPY_H2D = 0
HJB_COPY_D2H = 1
HJB_COPY_D2D = 2

HJB_MODEL_NONE = 0
HJB_MODEL_QUAT_EULER321 = 1
HJB_COST_DEFAULT, HJB_COST_F64 = 0, 1


class hjb_term(C.Structure):
    _fields_ = [("mask", C.c_uint32), ("reserved", C.c_uint32), ("data", C.c_void_p)]


class hjb_problem(C.Structure):
    _fields_ = [
        ("D", C.c_int32),
        ("C", C.c_int32),
        ("n", C.c_int32 * HJB_MAX_D),
        ("m", C.c_int32 * HJB_MAX_C),
        ("dtype", C.c_int32),
        ("index_base", C.c_int32),
        ("knots", C.POINTER(C.c_double) * HJB_MAX_D),
        ("n_next_terms", C.c_int32 * HJB_MAX_D),
        ("next_terms", (hjb_term * HJB_MAX_TERMS) * HJB_MAX_D),
        ("n_cost_terms", C.c_int32),
        ("idx_dtype", C.c_int32),
        ("cost_terms", hjb_term * HJB_MAX_TERMS),
        ("slab_begin", C.c_int32),
        ("slab_end", C.c_int32),
        ("halo_lo", C.c_int32),
        ("halo_hi", C.c_int32),
        ("model", C.c_int32),
        ("table_dtype", C.c_int32),
        ("model_h", C.c_double),
        ("model_tables", C.c_void_p * 4),
        ("cost_dtype", C.c_int32),
        ("reserved_", C.c_int32),
    ]


hjb_progress_fn = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double)


class hjb_probe(C.Structure):
    _fields_ = [
        ("lo", C.c_int32 * HJB_MAX_D),
        ("hi", C.c_int32 * HJB_MAX_D),
        ("control", C.c_int32 * HJB_MAX_C),
        ("reserved", C.c_int32),
        ("g", C.c_void_p),
        ("x_next", C.c_void_p),
        ("j_interp", C.c_void_p),
    ]


class hjb_solve_opts(C.Structure):
    _fields_ = [
        ("n_stages", C.c_int32),
        ("monitor_period", C.c_int32),
        ("monitor_tol", C.c_double),
        ("terminal", C.c_void_p),
        ("J_final", C.c_void_p),
        ("idx_final", C.c_void_p),
        ("J_stages", C.c_void_p),
        ("idx_stages", C.c_void_p),
        ("progress", hjb_progress_fn),
        ("progress_user", C.c_void_p),
        ("probe", C.POINTER(hjb_probe)),
        ("progress_every_stage", C.c_int32),
        ("monitor_single", C.c_int32),
    ]


class hjb_result(C.Structure):
    _fields_ = [
        ("stages_done", C.c_int32),
        ("stopped_early", C.c_int32),
        ("sweep_ms", C.c_double),
        ("last_e", C.c_double),
        ("last_e2", C.c_double),
    ]


class hjb_info(C.Structure):
    _fields_ = [
        ("n_states", C.c_int64),
        ("n_controls", C.c_int64),
        ("j_elems", C.c_int64),
        ("kernel_variant", C.c_int32),
        ("lds_bytes", C.c_int32),
        ("block", C.c_int32),
        ("grid", C.c_int32),
        ("halo_needed_lo", C.c_int32),
        ("halo_needed_hi", C.c_int32),
        ("idx_bytes", C.c_int32),
        ("table_dtype", C.c_int32),
        ("cost_dtype", C.c_int32),
        ("reserved_", C.c_int32),
    ]


# every symbol include/hjbdp.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "hjb_version": (C.c_char_p, []),
    "hjb_status_string": (C.c_char_p, [C.c_int32]),
    "hjb_device_count": (C.c_int32, []),
    "hjb_test_hook": (C.c_int32, [C.c_char_p, C.c_int64]),
    "hjb_create": (C.c_int32, [C.POINTER(hjb_problem), C.c_int32, C.POINTER(C.c_void_p)]),
    "hjb_destroy": (C.c_int32, [C.c_void_p]),
    "hjb_last_error": (C.c_char_p, [C.c_void_p]),
    "hjb_get_info": (C.c_int32, [C.c_void_p, C.POINTER(hjb_info)]),
    "hjb_set_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64]),
    "hjb_get_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "hjb_backup_stage": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hjb_backup_stage_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hjb_check_device_status": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "hjb_solve": (C.c_int32, [C.c_void_p, C.POINTER(hjb_solve_opts), C.POINTER(hjb_result)]),
    "hjb_solve_batch": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(hjb_solve_opts)), C.POINTER(C.POINTER(hjb_result))]),
    "hjb_policy_lookup": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                      C.POINTER(C.POINTER(C.c_double)), C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                      C.c_void_p]),
    # device-buffer helpers
    "hjb_device_malloc": (C.c_int32, [C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]),
    "hjb_device_free": (C.c_int32, [C.c_int32, C.c_void_p]),
    "hjb_device_mem_info": (C.c_int32, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hjb_device_copy": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "hjb_device_fill_separable": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "hjb_rank_fill_separable": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "hjb_device_gather": (C.c_int32, [C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_int64, C.c_void_p]),
    "hjb_probe_stage": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(hjb_probe)]),
    # flat builder API (primitives and plain arrays only: what MATLAB's calllib can marshal)
    "hjb_problem_new": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                    C.POINTER(C.c_void_p)]),
    "hjb_problem_set_knots": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.c_int32]),
    "hjb_problem_add_next_term": (C.c_int32, [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_int64]),
    "hjb_problem_add_cost_term": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64]),
    "hjb_problem_set_slab": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "hjb_problem_set_types": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "hjb_problem_set_cost_type": (C.c_int32, [C.c_void_p, C.c_int32]),
    "hjb_problem_set_model": (C.c_int32, [C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hjb_problem_permute_axes": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "hjb_problem_suggest_order": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hjb_create_from": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "hjb_problem_free": (C.c_int32, [C.c_void_p]),
    "hjb_problem_last_error": (C.c_char_p, [C.c_void_p]),
    "hjb_solve_flat": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "hjb_get_info_flat": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)]),
    # single-process multi-GPU sweep
    "hjb_create_multi": (C.c_int32, [C.POINTER(hjb_problem), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "hjb_create_multi_from": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "hjb_solve_multi": (C.c_int32, [C.c_void_p, C.POINTER(hjb_solve_opts), C.POINTER(hjb_result)]),
    "hjb_solve_multi_flat": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "hjb_multi_slab_info": (C.c_int32, [C.c_void_p, C.c_int32] + [C.POINTER(C.c_int32)] * 6),
    "hjb_multi_set_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64]),
    "hjb_destroy_multi": (C.c_int32, [C.c_void_p]),
    "hjb_multi_last_error": (C.c_char_p, [C.c_void_p]),
    # one process per GPU: a rank's slab as interior + boundary strips
    "hjb_rank_create": (C.c_int32, [C.POINTER(hjb_problem), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "hjb_rank_create_from": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "hjb_rank_comm_available": (C.c_int32, []),
    "hjb_rank_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "hjb_rank_comm_init": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "hjb_rank_comm_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hjb_rank_exchange": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hjb_rank_transfer_stream": (C.c_void_p, [C.c_void_p]),
    "hjb_rank_step": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hjb_rank_step_post": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hjb_rank_monitor_sums": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "hjb_rank_sweep": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "hjb_rank_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "hjb_rank_stage": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hjb_rank_stage_post": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hjb_rank_wait_strips": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "hjb_rank_set_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64]),
    "hjb_rank_get_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "hjb_rank_check_status": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "hjb_rank_destroy": (C.c_int32, [C.c_void_p]),
    "hjb_rank_last_error": (C.c_char_p, [C.c_void_p]),
}
HJB_LOOKUP_NEAREST = 0
HJB_LOOKUP_LINEAR = 1


def bind(lib, symbols=SYMBOLS):
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    return lib
