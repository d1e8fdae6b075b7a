// hjbdp_api.hip - the single-device C ABI of libhjbdp (include/hjbdp.h): hjb_create .. hjb_solve, options, probe block, policy lookup.
// gfx950 (MI355X) only; no CPU fallback - without a HIP device every compute entry point returns HJB_E_DEVICE.
#include "hjbdp_host.h"
#include "kernels_lookup.h"

using namespace hjbhost;

template <typename T>
int policy_lookup_t(int32_t D, const int32_t *n, const double *const *knots, const void *values, int64_t nq,
                           const void *queries, int32_t method, void *out) {
    Handle tmp;   // only for allocation bookkeeping and error text
    Handle *h = &tmp;
    DLookup L{};
    L.D = D;
    L.method = method;
    int64_t s = 1;
    for (int a = 0; a < D; ++a) {
        std::vector<T> kk(n[a]), rdx(n[a]);
        for (int i = 0; i < n[a]; ++i) kk[i] = (T)knots[a][i];
        for (int i = 0; i + 1 < n[a]; ++i) {
            if (!(kk[i + 1] > kk[i])) {
                for (void *d : h->allocs) (void)hipFree(d);      // the axes uploaded so far
                h->allocs.clear();
                h->arena_left = 0;
                g_last_error = "lookup: knots not strictly increasing";
                return HJB_E_INVALID;
            }
            rdx[i] = (T)1 / (T)(kk[i + 1] - kk[i]);
        }
        rdx[n[a] - 1] = (T)0;
        void *dk = nullptr, *dr = nullptr;
        int st = upload(h, kk, &dk);
        if (!st) st = upload(h, rdx, &dr);
        if (st) { for (void *d : h->allocs) (void)hipFree(d); return st; }
        L.knots[a] = dk;
        L.rdx[a] = dr;
        L.n[a] = n[a];
        L.stride[a] = s;
        s *= n[a];
    }
    void *dV = nullptr, *dQ = nullptr, *dO = nullptr;
    int st = dev_alloc(h, (size_t)s * sizeof(T), &dV);
    if (!st) st = dev_alloc(h, (size_t)nq * D * sizeof(T), &dQ);
    if (!st) st = dev_alloc(h, (size_t)nq * sizeof(T), &dO);
    hipError_t e = hipSuccess;
    if (!st) {
        e = hipMemcpy(dV, values, (size_t)s * sizeof(T), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dQ, queries, (size_t)nq * D * sizeof(T), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            const int grid = (int)std::min<int64_t>((nq + 255) / 256, 65536);
            dim3 g(std::max(grid, 1)), b(256);
            switch (D) {
                case 1: hipLaunchKernelGGL((k_policy_lookup<T, 1>), g, b, 0, nullptr, L, (const T *)dV, nq, (const T *)dQ, (T *)dO); break;
                case 2: hipLaunchKernelGGL((k_policy_lookup<T, 2>), g, b, 0, nullptr, L, (const T *)dV, nq, (const T *)dQ, (T *)dO); break;
                case 3: hipLaunchKernelGGL((k_policy_lookup<T, 3>), g, b, 0, nullptr, L, (const T *)dV, nq, (const T *)dQ, (T *)dO); break;
                case 4: hipLaunchKernelGGL((k_policy_lookup<T, 4>), g, b, 0, nullptr, L, (const T *)dV, nq, (const T *)dQ, (T *)dO); break;
                case 5: hipLaunchKernelGGL((k_policy_lookup<T, 5>), g, b, 0, nullptr, L, (const T *)dV, nq, (const T *)dQ, (T *)dO); break;
                default: hipLaunchKernelGGL((k_policy_lookup<T, 6>), g, b, 0, nullptr, L, (const T *)dV, nq, (const T *)dQ, (T *)dO); break;
            }
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpy(out, dO, (size_t)nq * sizeof(T), hipMemcpyDeviceToHost);
        }
    }
    for (void *d : h->allocs) (void)hipFree(d);
    h->allocs.clear();
    h->arena_left = 0;
    if (st) return st;
    if (e != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hjb_policy_lookup: %s", hipGetErrorString(e));
    return HJB_OK;
}

extern "C" {

const char *hjb_version(void) { return "hjbdp 0.1.0 (gfx950)"; }

// Fault injection for the test suite, by explicit call only (the environment never changes what the library does).
int32_t hjb_test_hook(const char *key, int64_t value) {
    if (!key) return fail(nullptr, HJB_E_INVALID, "hjb_test_hook: null key");
    if (!strcmp(key, "fail_tab64_scratch")) { g_test_fail_tab64_scratch.store(value != 0); return HJB_OK; }
    if (!strcmp(key, "fail_tabled_alloc")) { g_test_fail_tabled_alloc.store(value != 0); return HJB_OK; }
    if (!strcmp(key, "rccl_only_env")) { g_test_rccl_only_env.store(value != 0); return HJB_OK; }
    return fail(nullptr, HJB_E_INVALID, "hjb_test_hook: unknown key '%s'", key);
}

const char *hjb_status_string(int32_t s) {
    switch (s) {
        case HJB_OK: return "ok";
        case HJB_E_INVALID: return "invalid argument";
        case HJB_E_UNSUPPORTED: return "unsupported";
        case HJB_E_DEVICE: return "device error";
        case HJB_E_NOMEM: return "out of memory";
        case HJB_E_HALO: return "query outside slab halo";
        default: return "unknown status";
    }
}

int32_t hjb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *hjb_last_error(hjb_handle hh) {
    Handle *h = (Handle *)hh;
    return h ? h->err.c_str() : g_last_error.c_str();
}

// Everything hjb_create checks or derives WITHOUT touching a device: argument validation, the label width, and (on
// request) the halo the last axis' tables imply.  hjb_create_multi / hjb_rank_create partition on these numbers alone.
}  // extern "C"
int64_t hjbhost::first_nonfinite(const void *data, int64_t n, bool f64) {
    // x - x is 0 for finite x and NaN for inf / NaN: blocks of 4096 summed branch-free (vectorisable), the element found on a hit
    constexpr int64_t kBlk = 4096;
    for (int64_t b0 = 0; b0 < n; b0 += kBlk) {
        const int64_t b1 = std::min(n, b0 + kBlk);
        double acc = 0.0;
        if (f64) { const double *x = (const double *)data; for (int64_t i = b0; i < b1; ++i) acc += x[i] - x[i]; }
        else { const float *x = (const float *)data; for (int64_t i = b0; i < b1; ++i) acc += (double)(x[i] - x[i]); }
        if (acc != 0.0) {
            for (int64_t i = b0; i < b1; ++i)
                if (!std::isfinite(f64 ? ((const double *)data)[i] : (double)((const float *)data)[i])) return i;
        }
    }
    return -1;
}

int hjbhost::analyse_problem(const hjb_problem *p, int *idx_bytes_out, int64_t *n_states_out, int *halo_lo, int *halo_hi) {
    if (!p) return fail(nullptr, HJB_E_INVALID, "null argument");
    if (p->D < 1 || p->D > HJB_MAX_D) return fail(nullptr, HJB_E_UNSUPPORTED, "D=%d not in 1..%d", p->D, HJB_MAX_D);
    if (p->C < 1 || p->C > HJB_MAX_C) return fail(nullptr, HJB_E_UNSUPPORTED, "C=%d not in 1..%d", p->C, HJB_MAX_C);
    if (p->dtype != HJB_F32 && p->dtype != HJB_F64 && p->dtype != HJB_F16S) return fail(nullptr, HJB_E_UNSUPPORTED, "dtype %d", p->dtype);
    if (p->index_base != 0 && p->index_base != 1) return fail(nullptr, HJB_E_INVALID, "index_base must be 0 or 1");
    if (p->model != HJB_MODEL_NONE && p->model != HJB_MODEL_QUAT_EULER321) return fail(nullptr, HJB_E_INVALID, "model %d", p->model);
    if (p->model == HJB_MODEL_QUAT_EULER321) {
        if (p->D != 6 || p->C != 3 || p->dtype == HJB_F64)
            return fail(nullptr, HJB_E_UNSUPPORTED, "HJB_MODEL_QUAT_EULER321 needs D=6, C=3, float32 arithmetic");
        for (int i = 0; i < 4; ++i)
            if (!p->model_tables[i]) return fail(nullptr, HJB_E_INVALID, "model_tables[%d] is null", i);
    }
    if (p->idx_dtype < HJB_IDX_I32 || p->idx_dtype > HJB_IDX_AUTO) return fail(nullptr, HJB_E_INVALID, "idx_dtype %d", p->idx_dtype);
    if (p->table_dtype != HJB_TAB_DEFAULT && p->table_dtype != HJB_TAB_F64) return fail(nullptr, HJB_E_INVALID, "table_dtype %d", p->table_dtype);
    if (p->cost_dtype != HJB_COST_DEFAULT && p->cost_dtype != HJB_COST_F64) return fail(nullptr, HJB_E_INVALID, "cost_dtype %d", p->cost_dtype);
    if (p->cost_dtype == HJB_COST_F64 && (p->dtype == HJB_F64 || p->model))
        return fail(nullptr, HJB_E_INVALID, "cost_dtype HJB_COST_F64 is for float32 arithmetic without a state model (a float64 problem sums its cost in float64 anyway)");
    if (p->table_dtype == HJB_TAB_F64 && (p->dtype == HJB_F64 || p->model))
        return fail(nullptr, HJB_E_INVALID, "table_dtype HJB_TAB_F64 is for float32 arithmetic without a state model (a float64 problem is float64 throughout)");
    const int G = p->D + p->C;
    int64_t nS = 1, nU = 1;
    for (int a = 0; a < p->D; ++a)       // sizes first: everything below multiplies them
        if (p->n[a] < 2) return fail(nullptr, HJB_E_INVALID, "n[%d]=%d < 2", a, p->n[a]);
    for (int c = 0; c < p->C; ++c) {
        if (p->m[c] < 1) return fail(nullptr, HJB_E_INVALID, "m[%d]=%d < 1", c, p->m[c]);
        if (nU > ((int64_t)1 << 31) / p->m[c]) return fail(nullptr, HJB_E_UNSUPPORTED, "too many controls");
        nU *= p->m[c];
    }
    for (int a = 0; a < p->D; ++a) {
        if (!p->knots[a]) return fail(nullptr, HJB_E_INVALID, "knots[%d] is null", a);
        for (int i = 0; i < p->n[a]; ++i)
            if (!std::isfinite(p->knots[a][i])) return fail(nullptr, HJB_E_INVALID, "knots[%d][%d] is not finite", a, i);
        for (int i = 0; i + 1 < p->n[a]; ++i)
            if (!(p->knots[a][i + 1] > p->knots[a][i]))
                return fail(nullptr, HJB_E_INVALID, "knots[%d] not strictly increasing at %d", a, i);
        const bool model_axis = p->model == HJB_MODEL_QUAT_EULER321 && a < 3;
        if (model_axis ? p->n_next_terms[a] != 0 : (p->n_next_terms[a] < 1 || p->n_next_terms[a] > HJB_MAX_TERMS))
            return fail(nullptr, HJB_E_INVALID, "n_next_terms[%d]=%d", a, p->n_next_terms[a]);
        for (int k = 0; k < p->n_next_terms[a]; ++k) {
            const hjb_term &t = p->next_terms[a][k];
            if (!t.data || (t.mask >> G)) return fail(nullptr, HJB_E_INVALID, "next term %d of axis %d: bad mask/data", k, a);
            if (term_elems(p, t.mask) >= ((int64_t)1 << 31)) return fail(nullptr, HJB_E_UNSUPPORTED, "next term %d of axis %d has >= 2^31 elements", k, a);
            const int64_t bad = first_nonfinite(t.data, term_elems(p, t.mask), p->dtype == HJB_F64 || p->table_dtype == HJB_TAB_F64);
            if (bad >= 0) return fail(nullptr, HJB_E_INVALID, "next term %d of axis %d: element %lld is not finite", k, a, (long long)bad);
        }
        // (six axes of 2^15 points would wrap a 64-bit product to 0 and pass every later size check)
        if (nS > kMaxStates / p->n[a]) return fail(nullptr, HJB_E_UNSUPPORTED, "more than 2^40 grid points (axes 0..%d)", a);
        nS *= p->n[a];
    }
    if (p->model == HJB_MODEL_QUAT_EULER321) {
        for (int i = 0; i < 4; ++i) {
            const int64_t bad = first_nonfinite(p->model_tables[i], (int64_t)p->n[0] * p->n[1] * p->n[2], false);
            if (bad >= 0) return fail(nullptr, HJB_E_INVALID, "model_tables[%d]: element %lld is not finite", i, (long long)bad);
        }
        if (!std::isfinite(p->model_h)) return fail(nullptr, HJB_E_INVALID, "model_h is not finite");
    }
    if (nU >= (int64_t)1 << 31) return fail(nullptr, HJB_E_UNSUPPORTED, "too many controls");
    if (p->n_cost_terms < 1 || p->n_cost_terms > HJB_MAX_TERMS) return fail(nullptr, HJB_E_INVALID, "n_cost_terms=%d", p->n_cost_terms);
    for (int k = 0; k < p->n_cost_terms; ++k)
        if (!p->cost_terms[k].data || (p->cost_terms[k].mask >> G)) return fail(nullptr, HJB_E_INVALID, "cost term %d: bad mask/data", k);
    for (int k = 0; k < p->n_cost_terms; ++k) {
        if (term_elems(p, p->cost_terms[k].mask) >= ((int64_t)1 << 31)) return fail(nullptr, HJB_E_UNSUPPORTED, "cost term %d has >= 2^31 elements", k);
        const int64_t bad = first_nonfinite(p->cost_terms[k].data, term_elems(p, p->cost_terms[k].mask), p->dtype == HJB_F64 || p->cost_dtype == HJB_COST_F64);
        if (bad >= 0) return fail(nullptr, HJB_E_INVALID, "cost term %d: element %lld is not finite", k, (long long)bad);
    }
    if (p->slab_begin || p->slab_end || p->halo_lo || p->halo_hi) {
        const int nl = p->n[p->D - 1];
        if (p->slab_begin < 0 || p->slab_end > nl || p->slab_begin >= p->slab_end || p->halo_lo < 0 || p->halo_hi < 0 ||
            p->slab_begin - p->halo_lo < 0 || p->slab_end + p->halo_hi > nl)
            return fail(nullptr, HJB_E_INVALID, "bad slab [%d,%d) halo %d/%d on axis of %d planes", p->slab_begin,
                        p->slab_end, p->halo_lo, p->halo_hi, nl);
        if ((p->slab_end + p->halo_hi) - (p->slab_begin - p->halo_lo) < 2)
            return fail(nullptr, HJB_E_INVALID, "slab + halo must span at least 2 planes");
    }
    int idx_bytes = 4;
    {
        const int64_t top = nU - 1 + p->index_base;                       // the largest label
        if (p->idx_dtype == HJB_IDX_U8 || (p->idx_dtype == HJB_IDX_AUTO && top <= 255)) idx_bytes = 1;
        else if (p->idx_dtype == HJB_IDX_U16 || (p->idx_dtype == HJB_IDX_AUTO && top <= 65535)) idx_bytes = 2;
        if ((idx_bytes == 1 && top > 255) || (idx_bytes == 2 && top > 65535))
            return fail(nullptr, HJB_E_INVALID, "idx_dtype %d cannot hold the label %lld", p->idx_dtype, (long long)top);
    }
    if (idx_bytes_out) *idx_bytes_out = idx_bytes;
    if (n_states_out) *n_states_out = nS;
    if (halo_lo && halo_hi) {
        const bool tab64 = p->table_dtype == HJB_TAB_F64;
        if (p->n_next_terms[p->D - 1] < 1) { *halo_lo = *halo_hi = 0; }
        else halo_of_problem(p, tab64, halo_lo, halo_hi);
    }
    return HJB_OK;
}
extern "C" {

int32_t hjb_create(const hjb_problem *p, int32_t device, hjb_handle *out) {
    if (!p || !out) return fail(nullptr, HJB_E_INVALID, "null argument");
    *out = nullptr;
    int idx_bytes = 4;
    {
        const int ast = analyse_problem(p, &idx_bytes, nullptr, nullptr, nullptr);
        if (ast) return ast;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, HJB_E_DEVICE, "no HIP device visible (libhjbdp has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, HJB_E_INVALID, "device %d not in 0..%d", device, ndev - 1);
    Handle *h = new Handle();
    h->device = device;
    h->idx_bytes = idx_bytes;
    h->tab64 = p->table_dtype == HJB_TAB_F64;
    h->cost64 = p->cost_dtype == HJB_COST_F64;
    h->dtype = p->dtype;
    h->esz = p->dtype == HJB_F16S ? 2 : (p->dtype == HJB_F32 ? 4 : 8);
    h->prob = *p;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        int st = fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        delete h;
        return st;
    }
    std::shared_lock<std::shared_mutex> create_lk(g_capture_mu);
    int st = build_handle(h, p);
    if (st) {
        g_last_error = h->err;
        for (void *d : h->allocs) (void)hipFree(d);
        delete h;
        return st;
    }
    // pointers in the kept copy must not be dereferenced later
    for (int a = 0; a < HJB_MAX_D; ++a) {
        h->prob.knots[a] = nullptr;
        for (int k = 0; k < HJB_MAX_TERMS; ++k) h->prob.next_terms[a][k].data = nullptr;
    }
    for (int k = 0; k < HJB_MAX_TERMS; ++k) h->prob.cost_terms[k].data = nullptr;
    choose_launch(h);
    const bool tab64_bad = h->tab64 && (h->launch_status != HJB_OK || h->variant < 5);
    const bool cost64_bad = h->cost64 && (h->launch_status != HJB_OK || (h->variant != 5 && h->variant != 7));
    if (tab64_bad || cost64_bad) {
        // the caller asked for float64 queries / a float64 stage cost: a handle that cannot serve them is not handed out
        st = h->launch_status != HJB_OK ? h->launch_status : HJB_E_UNSUPPORTED;
        if (h->err.empty()) {
            if (tab64_bad) (void)fail(h, st, "table_dtype HJB_TAB_F64: the (cell, t) tables could not be built; table_dtype = HJB_TAB_DEFAULT (Python: table_dtype=None) runs float32 queries");
            else (void)fail(h, st, "cost_dtype HJB_COST_F64: the tables of the kernels that serve it (5, 7) could not be built (status %d); cost_dtype = HJB_COST_DEFAULT sums the cost terms in float32", h->launch_status);
        }
        g_last_error = h->err;
        if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
        for (void *d : h->allocs) (void)hipFree(d);
        delete h;
        return st;
    }
    *out = (hjb_handle)h;
    return HJB_OK;
}

int32_t hjb_destroy(hjb_handle hh) {
    Handle *h = (Handle *)hh;
    if (!h) return HJB_OK;
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    for (void *d : h->allocs) (void)hipFree(d);
    delete h;
    return HJB_OK;
}

int32_t hjb_get_info(hjb_handle hh, hjb_info *info) {
    Handle *h = (Handle *)hh;
    if (!h || !info) return fail(h, HJB_E_INVALID, "null argument");
    info->n_states = h->n_owned;
    info->n_controls = h->nU;
    info->j_elems = h->j_elems;
    info->kernel_variant = h->variant;
    info->lds_bytes = h->variant == 4 ? (int32_t)(uniwin_active(h) ? h->uw_lds : h->packed2_lds) : h->variant == 2 ? (int32_t)h->packed_lds
                      : (h->variant == 1 ? (int32_t)h->nested_lds
                      : (h->variant == 3 && h->split_j_in_lds ? (int32_t)(h->j_elems * h->esz) : 0));
    info->block = h->block;
    info->grid = h->grid;
    info->halo_needed_lo = h->halo_need_lo;
    info->halo_needed_hi = h->halo_need_hi;
    info->idx_bytes = h->idx_bytes;
    info->table_dtype = h->tab64 ? HJB_TAB_F64 : HJB_TAB_DEFAULT;
    info->cost_dtype = h->cost64 ? HJB_COST_F64 : HJB_COST_DEFAULT;
    info->reserved_ = 0;
    return HJB_OK;
}

int32_t hjb_set_option(hjb_handle hh, const char *key, int64_t value) {
    Handle *h = (Handle *)hh;
    if (!h || !key) return fail(h, HJB_E_INVALID, "null argument");
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);     // may build tables (allocation, device sync)
    {   // options that rewrite device-resident plans or tables IN PLACE wait for the whole device first: a stage of this handle may
        // still be in flight on a non-blocking stream - the handle's own, a rank's strip streams, a caller's.  Not hot.
        static const char *const kRewrites[] = {"prep_mfma", "cs_dpp", "cs_split", "cs_coop", "cs_xcd_axis", "cs_xcd_mod", "chunk_order",
                                                "uw_tile", "uw_block", "axis0_table", "window_planes", "uniwin"};
        for (const char *k : kRewrites)
            if (!strcmp(key, k)) {
                HIP_TRY(h, hipSetDevice(h->device));
                HIP_TRY(h, hipDeviceSynchronize());
                break;
            }
    }
    if (!strcmp(key, "variant")) {
        if (value < -1 || value > 7) return fail(h, HJB_E_INVALID, "variant %lld unknown", (long long)value);
        if (value == 7) {
            const int cst = ensure_colsweep(h);
            if (cst) return cst;
            if (h->cs_state != 1)
                return fail(h, HJB_E_UNSUPPORTED, "variant 7 (column sweep) needs D = 4, one control dim, float32 arithmetic, axes 0/1 "
                            "independent of the control (and of each other's state dim), axes 2/3 depending on state dims 2, 3 and the "
                            "control only, control terms of the cost involving the control only, and <= %d groups of corner rows per (i2, i3)", kCsGMax);
        }
        if (value == 6 && !h->row_ok)
            return fail(h, HJB_E_UNSUPPORTED, "variant 6 (one wave per grid row) needs D >= 2, per-axis tables that fit, and "
                        "no axis other than axis 0 depending on state dim 0");
        if (h->dtype == HJB_F16S && value >= 1 && value <= 3)
            return fail(h, HJB_E_UNSUPPORTED, "variant %lld does not support float16 J storage (use 0, 4, 5 or 6)", (long long)value);
        if (h->tab64 && value >= 0 && value <= 4)
            return fail(h, HJB_E_UNSUPPORTED, "variant %lld evaluates the next-state terms in the kernel, in float32; a problem with "
                        "table_dtype HJB_TAB_F64 runs on the table-driven kernels (5, 6, 7)", (long long)value);
        if (h->cost64 && value >= 0 && value != 5 && value != 7)
            return fail(h, HJB_E_UNSUPPORTED, "variant %lld sums the stage cost in float32; a problem with cost_dtype HJB_COST_F64 runs on "
                        "the tabled kernel (5) or the column sweep (7)", (long long)value);
        if (h->hp.model && value != -1 && value != 4)
            return fail(h, HJB_E_UNSUPPORTED, "a problem with a state model runs on variant 4 only");
        if (value == 5 && !h->tabled_ok)
            return fail(h, HJB_E_UNSUPPORTED, "variant 5 (tabled) needs per-axis tables that fit (see hjbdp.hip)");
        if (value == 4 && !h->packed_mode)
            return fail(h, HJB_E_UNSUPPORTED, "variant 4 (packed, control pairs) needs float32 and the canonical spacecraft structure");
        if (value == 2 && h->packed_mode != 1)
            return fail(h, HJB_E_UNSUPPORTED, "variant 2 (packed) needs float32 and the canonical spacecraft structure (see kernels_packed.h)");
        if (value == 2) { const int ast = ensure_axis0_table(h); if (ast) return ast; }     // variant 2 reads every axis from its table
        if (value == 1 && !h->nested_ok)
            return fail(h, HJB_E_UNSUPPORTED, "variant 1 (control-nested) needs: only the last state axis depends on the innermost control dim");
        h->forced_variant = (int)value;
        choose_launch(h);
        if (value >= 0 && h->variant != (int)value) {      // e.g. the tables of a forced variant 5/6 could not be built
            const int lst = h->launch_status != HJB_OK ? h->launch_status : HJB_E_UNSUPPORTED;
            h->forced_variant = -1;
            choose_launch(h);
            return fail(h, lst, "variant %lld could not be set up (%s); the automatic choice is in effect", (long long)value,
                        h->err.empty() ? "not applicable" : h->err.c_str());
        }
        return HJB_OK;
    }
    if (!strcmp(key, "prep_mfma")) {       // rebuild every (cell, weight) table: 1 = MFMA outer-sum form where it applies
        if (h->variant == 5 || h->variant == 6 || h->variant == 7) { const int tst = ensure_tabled(h); if (tst) return tst; }
        HIP_TRY(h, hipSetDevice(h->device));
        const int rst = rebuild_tables(h, value != 0);
        if (rst) return rst;
        if (h->cs_state == 1) {            // variant 7's plan is derived from the tables: same bits, nothing to redo
        }
        return HJB_OK;
    }
    if (!strcmp(key, "cs_dpp")) {                                    // 0: variant 7 loads both axis-0 neighbours (testing)
        h->cs_dpp = value != 0;
        if (h->cs_state == 1) {
            bool dok = false;
            const int cst = colsweep_dpp_ok_f32(h, &dok);
            if (cst) return cst;
            h->hcs.dpp = (dok && h->cs_dpp) ? 1 : 0;
            colsweep_split(h);
            { const int ust = colsweep_upload(h); if (ust) return ust; }
            choose_launch(h);
        }
        return HJB_OK;
    }
    if (!strcmp(key, "block")) {                                     // variant 3: threads per workgroup = 64 x the states a workgroup sweeps side by side
        if (h->variant != 3 || (value != 256 && value != 512 && value != 1024)) return fail(h, HJB_E_UNSUPPORTED, "block: 256, 512 or 1024 on kernel variant 3");
        if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
        h->block = (int)value;
        return HJB_OK;
    }
    if (!strcmp(key, "grid")) {                                      // workgroups per launch of the grid-stride stage kernels (timing experiments)
        if (value < 1 || value > (1 << 20)) return fail(h, HJB_E_INVALID, "%s out of range", key);
        if (h->variant == 7 || (h->variant == 4 && uniwin_active(h)) || h->variant == 1)
            return fail(h, HJB_E_UNSUPPORTED, "the launch size of kernel variant %d is part of its plan", h->variant);
        if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
        h->grid = (int)value;                                        // (until the next option that re-chooses the launch)
        return HJB_OK;
    }
    if (!strcmp(key, "cs_split")) {                                  // variant 7: parts a column is swept in (0 = automatic)
        if (value < 0 || value > 64) return fail(h, HJB_E_INVALID, "%s out of range", key);
        h->cs_split = (int)value;
        if (h->cs_state == 1) {
            colsweep_split(h);
            { const int ust = colsweep_upload(h); if (ust) return ust; }
            choose_launch(h);
        }
        return HJB_OK;
    }
    if (!strcmp(key, "tabled_i32")) {                                // 0: variant 5 in its 64-bit form whatever the sizes (A/B timing, tests)
        h->tabled_i32_on = value != 0;
        return HJB_OK;
    }
    if (!strcmp(key, "cs_coop")) {                                   // 0: variant 7 runs one wave per column (testing)
        h->cs_coop = value != 0;
        if (h->cs_state == 1) {
            h->hcs.coop = h->cs_coop ? h->cs_coop_epl : 0;
            { const int ust = colsweep_upload(h); if (ust) return ust; }
            choose_launch(h);
        }
        return HJB_OK;
    }
    if (!strcmp(key, "cs_xcd_axis")) {     // variant 7: which axis the XCDs split (0 = group axis, 1 = window axis)
        if (value < 0 || value > 1) return fail(h, HJB_E_INVALID, "%s out of range", key);
        h->cs_xcd_axis = (int)value;
        if (h->cs_state == 1) {
            std::vector<int32_t> plan((size_t)h->hp.n[2] * h->hp.n[3] * kCsPlanWords);
            HIP_TRY(h, hipMemcpy(plan.data(), h->hcs.plan, plan.size() * 4, hipMemcpyDeviceToHost));
            const int cst = colsweep_map(h, plan);
            if (cst) return cst;
            { const int ust = colsweep_upload(h); if (ust) return ust; }
            choose_launch(h);
        }
        return HJB_OK;
    }
    if (!strcmp(key, "cs_xcd_mod")) {      // variant 7: residue modulus of the column -> XCD assignment (0 = automatic)
        if (value < -1 || value > 4096) return fail(h, HJB_E_INVALID, "%s out of range", key);
        h->cs_xcd_mod = (int)value;
        if (h->cs_state == 1) {
            std::vector<int32_t> plan((size_t)h->hp.n[2] * h->hp.n[3] * kCsPlanWords);
            HIP_TRY(h, hipMemcpy(plan.data(), h->hcs.plan, plan.size() * 4, hipMemcpyDeviceToHost));
            const int cst = colsweep_map(h, plan);
            if (cst) return cst;
            { const int ust = colsweep_upload(h); if (ust) return ust; }
            choose_launch(h);
        }
        return HJB_OK;
    }
    if (!strcmp(key, "lds_pad")) {
        if (value < 0 || value > 128 * 1024) return fail(h, HJB_E_INVALID, "lds_pad out of range");
        h->lds_pad = (size_t)value;
        if (h->uniwin_ok) {          // K15 launches ONE generation of workgroups: its grid follows the occupancy
            uniwin_tiles(h);
            HIP_TRY(h, hipSetDevice(h->device));
            HIP_TRY(h, hipDeviceSynchronize());
            { DUniwin tmp = h->huw; if (!h->uw_claim) tmp.counters = nullptr; HIP_TRY(h, hipMemcpy(h->duw, &tmp, sizeof(DUniwin), hipMemcpyHostToDevice)); }
            choose_launch(h);
        }
        if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }   // the captured launches carry the old LDS size
        return HJB_OK;
    }
    if (!strcmp(key, "row_lean")) {
        h->row_lean = value != 0;
        if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }   // the captured launches are the other form
        return HJB_OK;
    }
    if (!strcmp(key, "temporal")) {
        if (value < 0 || value > 2) return fail(h, HJB_E_INVALID, "temporal must be 0, 1 or 2");
        h->use_temporal = (int)value;
        return HJB_OK;
    }
    if (!strcmp(key, "graph")) {
        h->use_graph = value != 0;
        return HJB_OK;
    }
    if (!strcmp(key, "axis0_table")) {      // 1: build the axis-0 (cell, t) table a mode-1 problem runs without (A/B timing, tests)
        if (value) { const int ast = ensure_axis0_table(h); if (ast) return ast; }
        return HJB_OK;
    }
    if (!strcmp(key, "window_planes")) {    // 4 / 3: variant 4's window modes with four planes (modes 2 / 3) or three (5 / 6)
        const bool three = h->packed_pre == 5 || h->packed_pre == 6, four = h->packed_pre == 2 || h->packed_pre == 3;
        if (!(three || four) || (value != 3 && value != 4)) return fail(h, HJB_E_UNSUPPORTED, "window_planes: 3 or 4, window modes only");
        if (value == 3 && !h->window3_ok) return fail(h, HJB_E_UNSUPPORTED, "window_planes 3: the inner control can skip a cell");
        if (value == 4 && three) { h->packed_pre -= 3; h->packed2_lds += 9 * 256 * 4 + 256 * 8; }
        if (value == 3 && four) { h->packed_pre += 3; h->packed2_lds -= 9 * 256 * 4 + 256 * 8; }
        choose_launch(h);          // (K15 serves the three-plane modes only: the grid follows; the captured launches are the other form)
        return HJB_OK;
    }
    if (!strcmp(key, "uniwin")) {           // K15 (kernels_uniwin.h): -1 automatic, 0 never, 1 whenever the structure holds
        if (value < -1 || value > 1) return fail(h, HJB_E_INVALID, "uniwin must be -1, 0 or 1");
        if (value == 1 && !h->uniwin_ok) return fail(h, HJB_E_UNSUPPORTED, "uniwin: the problem's rate axes are not shared by a chunk (kernels_uniwin.h)");
        h->uniwin_on = (int)value;
        choose_launch(h);
        return HJB_OK;
    }
    if (!strcmp(key, "uw_claim")) {         // K15: 1 = dynamic claim of the chunk walk's positions (default), 0 = fixed stride (A/B)
        if (!h->uniwin_ok) return fail(h, HJB_E_UNSUPPORTED, "uw_claim: K15 only");
        if (value != 0 && value != 1) return fail(h, HJB_E_INVALID, "uw_claim must be 0 or 1");
        static_assert(sizeof(void *) == 8, "");
        h->uw_claim = (int)value;
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipDeviceSynchronize());
        DUniwin tmp = h->huw;
        if (!h->uw_claim) tmp.counters = nullptr;
        HIP_TRY(h, hipMemcpy(h->duw, &tmp, sizeof(DUniwin), hipMemcpyHostToDevice));
        if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
        return HJB_OK;
    }
    if (!strcmp(key, "uw_tile") || !strcmp(key, "uw_block")) {
        // K15: log2 tile extents lA + 8 * lB + 64 * lC of the chunk walk (0: default) / states per chunk = threads per workgroup (256, 64)
        const bool tile = !strcmp(key, "uw_tile");
        if (tile ? (value < 0 || value > 511) : (value != 64 && value != 256)) return fail(h, HJB_E_INVALID, "%s out of range", key);
        if (!h->uniwin_ok) return fail(h, HJB_E_UNSUPPORTED, "%s: K15 only", key);
        if (tile) h->uw_tile = (int)value; else h->uw_block = (int)value;
        uniwin_tiles(h);
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipDeviceSynchronize());
        { DUniwin tmp = h->huw; if (!h->uw_claim) tmp.counters = nullptr; HIP_TRY(h, hipMemcpy(h->duw, &tmp, sizeof(DUniwin), hipMemcpyHostToDevice)); }
        choose_launch(h);
        return HJB_OK;
    }
    if (!strcmp(key, "chunk_order")) {      // variant 4, window modes: 0 transposed visiting order of the 256-state chunks, 1 state order
        if (value != 0 && value != 1) return fail(h, HJB_E_INVALID, "chunk_order must be 0 or 1");
        if (!h->dn) return fail(h, HJB_E_UNSUPPORTED, "chunk_order: variant 4's window modes only");
        h->hn.chunk_order = (int32_t)value;
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipMemcpy(h->dn, &h->hn, sizeof(DNested), hipMemcpyHostToDevice));
        return HJB_OK;
    }
    if (!strcmp(key, "monitor_single")) {   // hjb_solve_opts.monitor_single for callers of the flat API (hjb_solve_flat)
        h->monitor_single = value != 0;
        return HJB_OK;
    }
    return fail(h, HJB_E_INVALID, "unknown option '%s'", key);
}

int32_t hjb_get_option(hjb_handle hh, const char *key, int64_t *value) {
    Handle *h = (Handle *)hh;
    if (!h || !key || !value) return fail(h, HJB_E_INVALID, "null argument");
    if (!strcmp(key, "variant")) *value = h->variant;
    else if (!strcmp(key, "graph")) *value = h->use_graph ? 1 : 0;
    else if (!strcmp(key, "axis0_table")) *value = h->axis0_inline ? 0 : 1;       // 0: mode 1 forms axis 0's (cell, t) in the kernel
    else if (!strcmp(key, "monitor_single")) *value = h->monitor_single ? 1 : 0;
    // variant 4's contraction mode (kernels_packed2.h; 7 / 8: K15, kernels_uniwin.h), -1: not eligible
    else if (!strcmp(key, "packed2_mode")) *value = h->packed_mode ? (uniwin_active(h) ? h->packed_pre + 2 : h->packed_pre) : -1;
    else if (!strcmp(key, "uniwin")) *value = uniwin_active(h) ? 1 : 0;               // the form in effect
    else if (!strcmp(key, "uniwin_ok")) *value = h->uniwin_ok ? 1 : 0;
    else if (!strcmp(key, "uw_block")) *value = h->uniwin_ok ? h->huw.block : 0;
    else if (!strcmp(key, "uw_claim")) *value = h->uniwin_ok ? h->uw_claim : 0;
    else if (!strcmp(key, "uniwin_slow_points")) *value = h->uniwin_ok ? h->uniwin_slow : -1;
    else if (!strcmp(key, "grid")) *value = h->grid;
    else if (!strcmp(key, "idx_bytes")) *value = h->idx_bytes;
    else if (!strcmp(key, "temporal")) *value = h->use_temporal;
    else if (!strcmp(key, "chunk_order")) *value = h->dn ? h->hn.chunk_order : 0;
    else if (!strcmp(key, "row_lean")) *value = h->row_lean ? 1 : 0;
    else if (!strcmp(key, "lds_pad")) *value = (int64_t)h->lds_pad;
    else if (!strcmp(key, "cs_xcd_mod")) *value = h->cs_xcd_mod;
    else if (!strcmp(key, "cs_xcd_axis")) *value = h->cs_xcd_axis;
    else if (!strcmp(key, "cs_split")) *value = h->variant == 7 ? h->hcs.split : 0;       // the value in effect
    else if (!strcmp(key, "tabled_i32")) *value = (h->tabled_i32 && h->tabled_i32_on) ? 1 : 0;      // the form variant 5 would run
    else if (!strcmp(key, "cs_coop_why")) *value = h->cs_coop_why;
    else if (!strcmp(key, "cs_rows")) *value = h->variant == 7 ? h->cs_rows_mid : 0;
    else if (!strcmp(key, "prep_mfma")) *value = h->prep_mfma;
    else if (!strcmp(key, "prep_mfma_tables")) *value = h->prep_mfma_axes;
    else if (!strcmp(key, "prep_tables")) *value = (int64_t)h->preps.size();
    else if (!strcmp(key, "prep_ns")) *value = (int64_t)(h->prep_us * 1e3);          // device time of the last table rebuild
    else if (!strcmp(key, "table_hash")) {
        uint64_t hv = 0;
        std::shared_lock<std::shared_mutex> lk(g_capture_mu);
        HIP_TRY(h, hipSetDevice(h->device));
        const int hst = table_hash(h, &hv);
        if (hst) return hst;
        *value = (int64_t)(hv & 0x7fffffffffffffffull);
    }
    else if (!strcmp(key, "cs_dpp")) *value = (h->variant == 7 && h->hcs.dpp) ? 1 : 0;          // the form in effect
    else if (!strcmp(key, "cs_coop")) *value = (h->variant == 7 && h->hcs.coop && h->cc_grid > 0) ? 1 : 0;   // the form in effect
    else if (!strcmp(key, "cs_groups")) *value = h->variant == 7 ? h->hcs.ng : 0;
    else if (!strcmp(key, "cs_group_axis")) *value = h->variant == 7 ? h->hcs.gax : -1;
    else return fail(h, HJB_E_INVALID, "unknown option '%s'", key);
    return HJB_OK;
}

int32_t hjb_backup_stage_device(hjb_handle hh, const void *dJ_next, void *dJ_out, void *d_idx_out, void *stream) {
    Handle *h = (Handle *)hh;
    if (!h || !dJ_next || !dJ_out) return fail(h, HJB_E_INVALID, "null argument");
    if (dJ_next == dJ_out) return fail(h, HJB_E_INVALID, "J_next and J_out must not alias");
    HIP_TRY(h, hipSetDevice(h->device));
    return launch_stage(h, dJ_next, dJ_out, d_idx_out, (hipStream_t)stream);
}

int32_t hjb_check_device_status(hjb_handle hh, void *stream) {
    Handle *h = (Handle *)hh;
    if (!h) return fail(h, HJB_E_INVALID, "null handle");
    HIP_TRY(h, hipSetDevice(h->device));
    return check_status(h, (hipStream_t)stream);
}

int32_t hjb_backup_stage(hjb_handle hh, const void *J_next, void *J_out, void *idx_out) {
    Handle *h = (Handle *)hh;
    if (!h || !J_next || !J_out) return fail(h, HJB_E_INVALID, "null argument");
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    HIP_TRY(h, hipSetDevice(h->device));
    int st = ensure_work(h);
    if (st) return st;
    const size_t jb = (size_t)h->j_elems * h->esz;
    HIP_TRY(h, hipMemcpy(h->dJ[0], J_next, jb, hipMemcpyHostToDevice));
    // keep halo planes of the output defined: start from the input
    HIP_TRY(h, hipMemcpy(h->dJ[1], h->dJ[0], jb, hipMemcpyDeviceToDevice));
    st = launch_stage(h, h->dJ[0], h->dJ[1], h->d_idx, nullptr);
    if (st) return st;
    st = check_status(h, nullptr);
    if (st) return st;
    HIP_TRY(h, hipMemcpy(J_out, h->dJ[1], jb, hipMemcpyDeviceToHost));
    if (idx_out) HIP_TRY(h, hipMemcpy(idx_out, h->d_idx, (size_t)h->n_owned * h->idx_bytes, hipMemcpyDeviceToHost));
    return HJB_OK;
}

int32_t hjb_solve(hjb_handle hh, const hjb_solve_opts *o, hjb_result *res) {
    Handle *h = (Handle *)hh;
    if (!h || !o) return fail(h, HJB_E_INVALID, "null argument");
    if (o->n_stages < 1) return fail(h, HJB_E_INVALID, "n_stages=%d", o->n_stages);
    if (h->j_elems != h->n_owned)
        return fail(h, HJB_E_UNSUPPORTED, "hjb_solve runs whole grids; drive slabs with hjb_backup_stage_device + a halo exchange");
    HIP_TRY(h, hipSetDevice(h->device));
    std::shared_lock<std::shared_mutex> unsafe_lk(g_capture_mu);    // allocation, synchronous copies, device sync
    int st = ensure_work(h);
    if (st) return st;
    const int64_t nS = h->n_owned;
    const size_t jb = (size_t)nS * h->esz;
    if (!h->stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    hipStream_t stream = h->stream;
    // optional per-stage capture: kernels write straight into the stage planes
    char *dJst = nullptr;
    char *dIst = nullptr;
    const size_t ib = (size_t)nS * h->idx_bytes;      // bytes of one plane of labels
    if (o->J_stages) {
        void *d = nullptr;
        if (hipMalloc(&d, jb * o->n_stages) != hipSuccess) return fail(h, HJB_E_NOMEM, "cannot hold %d J stages on the device", o->n_stages);
        dJst = (char *)d;
        if (hipMemset(dJst, 0, jb * o->n_stages) != hipSuccess) { (void)hipFree(dJst); return fail(h, HJB_E_DEVICE, "hipMemset of the J stage planes failed"); }
    }
    if (o->idx_stages) {
        void *d = nullptr;
        if (hipMalloc(&d, ib * o->n_stages) != hipSuccess) {
            if (dJst) (void)hipFree(dJst);
            return fail(h, HJB_E_NOMEM, "cannot hold %d idx stages on the device", o->n_stages);
        }
        dIst = (char *)d;
        if (hipMemset(dIst, 0, ib * o->n_stages) != hipSuccess) {
            if (dJst) (void)hipFree(dJst);
            (void)hipFree(dIst);
            return fail(h, HJB_E_DEVICE, "hipMemset of the idx stage planes failed");
        }
    }
    // optional probe block (the reference's debug taps): one plane of each requested output per stage
    DProbe pr{};
    char *dPg = nullptr, *dPx = nullptr, *dPj = nullptr;
    const size_t tsz = h->dtype == HJB_F64 ? 8 : 4;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto cleanup = [&]() {
        // hipFree is one of the calls a stream capture elsewhere in the process must not see: hold the shared lock
        std::shared_lock<std::shared_mutex> lk(g_capture_mu, std::defer_lock);
        if (!unsafe_lk.owns_lock()) lk.lock();
        if (ev0) { (void)hipEventDestroy(ev0); ev0 = nullptr; }
        if (ev1) { (void)hipEventDestroy(ev1); ev1 = nullptr; }
        if (dJst) { (void)hipFree(dJst); dJst = nullptr; }
        if (dIst) { (void)hipFree(dIst); dIst = nullptr; }
        if (dPg) { (void)hipFree(dPg); dPg = nullptr; }
        if (dPx) { (void)hipFree(dPx); dPx = nullptr; }
        if (dPj) { (void)hipFree(dPj); dPj = nullptr; }
    };
    if (o->probe) {
        st = make_probe(h, o->probe, &pr);
        if (st) { cleanup(); return st; }
        const size_t pb = (size_t)pr.B * tsz;
        void *d = nullptr;
        // zero-filled: planes of stages an early stop never runs come back as zeros, like J_stages / idx_stages
        auto grab = [&](size_t bytes, char **out) {
            if (hipMalloc(&d, bytes) != hipSuccess) return false;
            *out = (char *)d;
            return hipMemset(d, 0, bytes) == hipSuccess;
        };
        if (o->probe->g && !grab(pb * o->n_stages, &dPg)) { cleanup(); return fail(h, HJB_E_NOMEM, "probe buffers"); }
        if (o->probe->x_next && !grab(pb * h->hp.D * o->n_stages, &dPx)) { cleanup(); return fail(h, HJB_E_NOMEM, "probe buffers"); }
        if (o->probe->j_interp && !grab(pb * o->n_stages, &dPj)) { cleanup(); return fail(h, HJB_E_NOMEM, "probe buffers"); }
    }
    const bool every_stage = o->progress && o->progress_every_stage;
#define SOLVE_TRY(expr)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            cleanup();                                                                             \
            return fail(h, HJB_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));           \
        }                                                                                          \
    } while (0)
    if (o->terminal) SOLVE_TRY(hipMemcpy(h->dJ[0], o->terminal, jb, hipMemcpyHostToDevice));
    else SOLVE_TRY(hipMemset(h->dJ[0], 0, jb));
    SOLVE_TRY(sync_setup());             // the sweep runs on the handle's own stream from here (other handles' sweeps are not waited for)
    // launch-bound sweeps: replay kGraphStages ping-pong launches per hipGraphLaunch
    const bool graph_ok = h->use_graph && !dJst && !dIst && !o->probe && !every_stage && o->n_stages >= 2 * kGraphStages;
    // K9: several stages per launch for local 2-D problems (no per-stage outputs, no monitor read-backs)
    bool tiled = false;
    if (h->use_temporal && !h->cost64 && !dJst && !dIst && !o->probe && !every_stage && o->monitor_period <= 0 && o->n_stages >= 2 * kTileK && h->forced_variant < 0) {
        if (h->tile2d < 0) {
            const int tst = examine_tile2d(h);
            if (tst) { cleanup(); return tst; }
        }
        tiled = h->tile2d == 1;
    }
    if (h->use_temporal == 2 && !tiled) {
        cleanup();
        return fail(h, HJB_E_UNSUPPORTED, "option temporal=2: several stages per launch do not apply (needs D=2, whole grid, "
                    "every query within one cell of its state, no per-stage outputs or monitor, >= %d stages)", 2 * kTileK);
    }
    if (h->gexec && h->gexec_tiled != tiled) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    unsafe_lk.unlock();
    if (graph_ok && !h->gexec) {
        std::unique_lock<std::shared_mutex> capture_lk(g_capture_mu);
        hipGraph_t graph = nullptr;
        SOLVE_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        int cst = HJB_OK;
        if (tiled) {          // kGraphStages = 4 launches of kTileK stages, ending in dJ[0]
            static_assert(kGraphStages % (2 * kTileK) == 0, "a graph must hold an even number of tile launches");
            for (int i = 0; i < kGraphStages / (2 * kTileK) && cst == HJB_OK; ++i) {
                cst = launch_tile2d(h, h->dJ[0], h->dJ[1], h->d_idx, kTileK, stream);
                if (cst == HJB_OK) cst = launch_tile2d(h, h->dJ[1], h->dJ[0], h->d_idx, kTileK, stream);
            }
        } else {
            for (int i = 0; i < kGraphStages / 2 && cst == HJB_OK; ++i) {
                cst = launch_stage(h, h->dJ[0], h->dJ[1], h->d_idx, stream);
                if (cst == HJB_OK) cst = launch_stage(h, h->dJ[1], h->dJ[0], h->d_idx, stream);
            }
        }
        h->gexec_tiled = tiled;
        hipError_t ce = hipStreamEndCapture(stream, &graph);
        if (cst != HJB_OK || ce != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            cleanup();
            return fail(h, HJB_E_DEVICE, "stage-loop graph capture failed: %s", hipGetErrorString(ce));
        }
        ce = hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ce != hipSuccess) { cleanup(); return fail(h, HJB_E_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(ce)); }
    }
    SOLVE_TRY(hipEventCreate(&ev0));
    SOLVE_TRY(hipEventCreate(&ev1));
    SOLVE_TRY(hipEventRecord(ev0, stream));
    const void *cur = h->dJ[0];
    int pp = 1;  // next ping-pong target
    const char *cur_idx = h->d_idx;
    int done = 0, early = 0;
    double fprev = 0, iprev = 0, e = 0, e2 = 0;
    int k_s = o->n_stages;
    while (k_s >= 1) {
        // stages up to and including the next monitor point (or all of them)
        int stop = 1;
        if (o->monitor_period > 0) stop = std::max(1, (k_s / o->monitor_period) * o->monitor_period);
        int run = k_s - stop + 1;
        if (graph_ok && run >= kGraphStages) {
            if (pp == 0) {   // make dJ[0] the current buffer: one eager stage
                st = launch_stage(h, cur, h->dJ[pp], h->d_idx, stream);
                if (st) { cleanup(); return st; }
                cur = h->dJ[pp]; pp ^= 1; ++done; --run; --k_s;
            }
            while (run >= kGraphStages) {
                SOLVE_TRY(hipGraphLaunch(h->gexec, stream));
                done += kGraphStages; run -= kGraphStages; k_s -= kGraphStages;
            }
        }
        while (tiled && run > 0) {                           // K9: up to kTileK stages per launch
            const int K = std::min(run, kTileK);
            st = launch_tile2d(h, cur, h->dJ[pp], h->d_idx, K, stream);
            if (st) { cleanup(); return st; }
            cur = h->dJ[pp];
            cur_idx = h->d_idx;
            pp ^= 1;
            done += K; run -= K; k_s -= K;
        }
        for (; run > 0; --run, --k_s) {
            void *outJ = dJst ? (void *)(dJst + (size_t)(k_s - 1) * jb) : h->dJ[pp];
            char *outI = dIst ? dIst + (size_t)(k_s - 1) * ib : h->d_idx;
            if (o->probe) {                              // taps of stage k_s: tables at the block, J_{k+1} = cur
                const size_t pb = (size_t)pr.B * tsz;
                pr.g = dPg ? dPg + (size_t)(k_s - 1) * pb : nullptr;
                pr.x_next = dPx ? dPx + (size_t)(k_s - 1) * pb * h->hp.D : nullptr;
                pr.j_interp = dPj ? dPj + (size_t)(k_s - 1) * pb : nullptr;
                st = launch_probe(h, pr, cur, stream);
                if (st) { cleanup(); return st; }
            }
            st = launch_stage(h, cur, outJ, outI, stream);
            if (st) { cleanup(); return st; }
            if (every_stage && !(o->monitor_period > 0 && k_s == stop)) {   // Dynamic_Solver.m:101: one line per stage
                float ems = 0;
                (void)hipEventRecord(ev1, stream);
                (void)hipEventSynchronize(ev1);
                (void)hipEventElapsedTime(&ems, ev0, ev1);
                o->progress(o->progress_user, k_s, 0.0, 0.0, ems * 1e-3);
            }
            cur = outJ;
            cur_idx = outI;
            if (!dJst) pp ^= 1;
            ++done;
        }
        // here k_s == stop - 1; the stage just computed has reference index `stop`
        if (o->monitor_period > 0 && (stop % o->monitor_period) == 0) {
            // Solver_pos_att.m:273-285: fsum50 = sum(F.Values(:)), idsum50 = sum(U_Optimal_id(:))
            st = launch_monitor_sums(h->dtype, o->monitor_single != 0 || h->monitor_single, cur, cur_idx, h->idx_bytes, nS, h->d_partials, h->d_sums, stream);
            if (st != HJB_OK) { cleanup(); return fail(h, HJB_E_DEVICE, "monitor reduction launch failed"); }
            double sums[2];
            unsafe_lk.lock();        // the handle's one lock (cleanup() on a failure below sees it held: no second shared lock)
            SOLVE_TRY(hipMemcpyAsync(sums, h->d_sums, sizeof sums, hipMemcpyDeviceToHost, stream));
            SOLVE_TRY(hipStreamSynchronize(stream));
            unsafe_lk.unlock();
            // Solver_pos_att.m:276-282: with a single fsum50, `e = fsum50 - fsum50_prev` is a single-precision subtraction
            // and `abs(e) < tol` compares in single (MATLAB casts the double tol); otherwise everything is double
            const bool msingle = (o->monitor_single != 0 || h->monitor_single) && h->dtype != HJB_F64;
            e = msingle ? (double)((float)sums[0] - (float)fprev) : sums[0] - fprev;
            e2 = sums[1] - iprev;
            fprev = sums[0];
            iprev = sums[1];
            if (o->progress) {
                float ms = 0;
                (void)hipEventRecord(ev1, stream);
                (void)hipEventSynchronize(ev1);
                (void)hipEventElapsedTime(&ms, ev0, ev1);
                o->progress(o->progress_user, stop, e, e2, ms * 1e-3);
            }
            if (msingle ? (std::fabs((float)e) < (float)o->monitor_tol) : (std::fabs(e) < o->monitor_tol)) { early = 1; break; }
        }
    }
    SOLVE_TRY(hipEventRecord(ev1, stream));
    SOLVE_TRY(hipEventSynchronize(ev1));
    float ms = 0;
    SOLVE_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    unsafe_lk.lock();
    st = check_status(h, stream);
    if (st) { cleanup(); return st; }
    if (o->J_final) SOLVE_TRY(hipMemcpy(o->J_final, cur, jb, hipMemcpyDeviceToHost));
    if (o->idx_final) SOLVE_TRY(hipMemcpy(o->idx_final, cur_idx, ib, hipMemcpyDeviceToHost));
    if (dPg) SOLVE_TRY(hipMemcpy(o->probe->g, dPg, (size_t)pr.B * tsz * o->n_stages, hipMemcpyDeviceToHost));
    if (dPx) SOLVE_TRY(hipMemcpy(o->probe->x_next, dPx, (size_t)pr.B * tsz * h->hp.D * o->n_stages, hipMemcpyDeviceToHost));
    if (dPj) SOLVE_TRY(hipMemcpy(o->probe->j_interp, dPj, (size_t)pr.B * tsz * o->n_stages, hipMemcpyDeviceToHost));
    if (o->J_stages) SOLVE_TRY(hipMemcpy(o->J_stages, dJst, jb * o->n_stages, hipMemcpyDeviceToHost));
    if (o->idx_stages) SOLVE_TRY(hipMemcpy(o->idx_stages, dIst, ib * o->n_stages, hipMemcpyDeviceToHost));
    cleanup();
    if (res) {
        res->stages_done = done;
        res->stopped_early = early;
        res->sweep_ms = ms;
        res->last_e = e;
        res->last_e2 = e2;
    }
    return HJB_OK;
#undef SOLVE_TRY
}

int32_t hjb_policy_lookup(int32_t device, int32_t dtype, int32_t D, const int32_t *n, const double *const *knots,
                          const void *values, int64_t nq, const void *queries, int32_t method, void *out) {
    if (!n || !knots || !values || !queries || !out) return fail(nullptr, HJB_E_INVALID, "null argument");
    if (D < 1 || D > HJB_MAX_D) return fail(nullptr, HJB_E_UNSUPPORTED, "D=%d", D);
    if (dtype != HJB_F32 && dtype != HJB_F64) return fail(nullptr, HJB_E_UNSUPPORTED, "dtype %d", dtype);
    if (method != HJB_LOOKUP_NEAREST && method != HJB_LOOKUP_LINEAR) return fail(nullptr, HJB_E_INVALID, "method %d", method);
    if (nq < 0) return fail(nullptr, HJB_E_INVALID, "nq < 0");
    for (int a = 0; a < D; ++a)
        if (n[a] < 2 || !knots[a]) return fail(nullptr, HJB_E_INVALID, "axis %d: need >= 2 knots", a);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, HJB_E_DEVICE, "no HIP device visible (libhjbdp has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, HJB_E_INVALID, "device %d", device);
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice failed");
    if (nq == 0) return HJB_OK;
    return dtype == HJB_F32 ? policy_lookup_t<float>(D, n, knots, values, nq, queries, method, out)
                            : policy_lookup_t<double>(D, n, knots, values, nq, queries, method, out);
}

int32_t hjb_probe_stage(hjb_handle hh, const void *J_next, const hjb_probe *probe) {
    Handle *h = (Handle *)hh;
    if (!h || !probe) return fail(h, HJB_E_INVALID, "null argument");
    if (probe->j_interp && !J_next) return fail(h, HJB_E_INVALID, "j_interp needs J_next");
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    HIP_TRY(h, hipSetDevice(h->device));
    DProbe pr{};
    int st = make_probe(h, probe, &pr);
    if (st) return st;
    const size_t tsz = h->dtype == HJB_F64 ? 8 : 4, pb = (size_t)pr.B * tsz;
    void *dg = nullptr, *dx = nullptr, *dj = nullptr;
    auto release = [&]() { if (dg) (void)hipFree(dg); if (dx) (void)hipFree(dx); if (dj) (void)hipFree(dj); };
    if ((probe->g && hipMalloc(&dg, pb) != hipSuccess) || (probe->x_next && hipMalloc(&dx, pb * h->hp.D) != hipSuccess) ||
        (probe->j_interp && hipMalloc(&dj, pb) != hipSuccess)) {
        release();
        return fail(h, HJB_E_NOMEM, "probe buffers");
    }
    pr.g = dg; pr.x_next = dx; pr.j_interp = dj;
    const void *dJn = nullptr;
    if (probe->j_interp) {
        st = ensure_work(h);
        if (st) { release(); return st; }
        if (hipMemcpy(h->dJ[0], J_next, (size_t)h->j_elems * h->esz, hipMemcpyHostToDevice) != hipSuccess) { release(); return fail(h, HJB_E_DEVICE, "copy of J_next failed"); }
        dJn = h->dJ[0];
    }
    st = launch_probe(h, pr, dJn, nullptr);
    if (!st) st = check_status(h, nullptr);
    hipError_t e = hipSuccess;
    if (!st && dg) e = hipMemcpy(probe->g, dg, pb, hipMemcpyDeviceToHost);
    if (!st && e == hipSuccess && dx) e = hipMemcpy(probe->x_next, dx, pb * h->hp.D, hipMemcpyDeviceToHost);
    if (!st && e == hipSuccess && dj) e = hipMemcpy(probe->j_interp, dj, pb, hipMemcpyDeviceToHost);
    release();
    if (st) return st;
    if (e != hipSuccess) return fail(h, HJB_E_DEVICE, "hjb_probe_stage: %s", hipGetErrorString(e));
    return HJB_OK;
}

}  // extern "C"
