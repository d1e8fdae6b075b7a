// hjbdp_setup.hip - libhjbdp host side: problem upload, stage-invariant tables, kernel choice and plans, the stage launch.
// gfx950 (MI355X) only; no CPU fallback - without a HIP device every compute entry point returns HJB_E_DEVICE.
#include "hjbdp_host.h"
#include "kernels_prep_mfma.h"

namespace hjbhost {

thread_local std::string g_last_error;
std::atomic<int> g_test_fail_tab64_scratch{0};
std::atomic<int> g_test_fail_tabled_alloc{0};
std::atomic<int> g_test_rccl_only_env{0};
std::shared_mutex g_capture_mu;

int fail(Handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    g_last_error = buf;
    return code;
}

// Every device allocation of a handle.  Small ones (plans, tables, cost terms, records: dozens per handle) are carved out of 1 MiB
// chunks: hipMalloc / hipFree cost 50 - 200 us each and hipFree waits for the device, which was 8 of the 64 ms of a whole
// Solver_pos_att.simplified_run (four handles; profiles/r06_batch_split.log, block 3).  256-byte aligned (the widest access of the
// kernels is 64 bytes: s_load_dwordx16 of a launch record).
int dev_alloc(Handle *h, size_t bytes, void **out) {
    constexpr size_t kChunk = (size_t)1 << 20, kSmall = (size_t)128 << 10, kAlign = 256;
    bytes = std::max<size_t>(bytes, 16);
    if (bytes <= kSmall) {
        const size_t need = (bytes + kAlign - 1) & ~(kAlign - 1);
        if (h->arena_left < need) {
            void *c = nullptr;
            HIP_TRY(h, hipMalloc(&c, kChunk));
            h->allocs.push_back(c);
            h->arena = (char *)c;
            h->arena_left = kChunk;
        }
        *out = h->arena;
        h->arena += need;
        h->arena_left -= need;
        return HJB_OK;
    }
    void *d = nullptr;
    HIP_TRY(h, hipMalloc(&d, bytes));
    h->allocs.push_back(d);
    *out = d;
    return HJB_OK;
}

int64_t term_elems(const hjb_problem *p, uint32_t mask) {
    int64_t s = 1;
    for (int d = 0; d < p->D + p->C; ++d)
        if (mask & (1u << d)) {
            const int64_t k = (d < p->D) ? p->n[d] : p->m[d - p->D];
            if (k < 1 || s > (INT64_MAX >> 1) / k) return INT64_MAX;      // saturates: every caller compares with a limit
            s *= k;
        }
    return s;
}

// upload one term, fill strides
template <typename T, typename TS = T>      // T: element type on the device, TS: element type of the caller's array
int make_term(Handle *h, const hjb_problem *p, const hjb_term &t, DTerm *out) {
    const int G = p->D + p->C;
    int64_t s = 1;
    for (int d = 0; d < HJB_MAX_G; ++d) out->stride[d] = 0;
    for (int d = 0; d < G; ++d) {
        if (t.mask & (1u << d)) {
            out->stride[d] = (int32_t)s;
            s *= (d < p->D) ? p->n[d] : p->m[d - p->D];
        }
    }
    std::vector<T> host((size_t)s);
    for (int64_t i = 0; i < s; ++i) host[(size_t)i] = (T)((const TS *)t.data)[i];
    void *d = nullptr;
    int st = upload(h, host, &d);
    if (st) return st;
    out->data = d;
    out->pad = 0;
    return HJB_OK;
}

// conservative range of an ordered term sum for a fixed index along `dim`
// (used for the halo the last axis needs)
template <typename T>
void term_minmax_along(const hjb_problem *p, const hjb_term &t, int dim, std::vector<double> &lo,
                       std::vector<double> &hi) {
    const int G = p->D + p->C;
    const int nd = p->n[dim];
    std::vector<double> tlo(nd, INFINITY), thi(nd, -INFINITY);
    int64_t total = term_elems(p, t.mask);
    int64_t stride_dim = 0, s = 1;
    for (int d = 0; d < G; ++d) {
        if (t.mask & (1u << d)) {
            if (d == dim) stride_dim = s;
            s *= (d < p->D) ? p->n[d] : p->m[d - p->D];
        }
    }
    const T *data = (const T *)t.data;
    if (!(t.mask & (1u << dim))) {
        double mn = INFINITY, mx = -INFINITY;
        for (int64_t i = 0; i < total; ++i) { mn = std::min(mn, (double)data[i]); mx = std::max(mx, (double)data[i]); }
        for (int i = 0; i < nd; ++i) { tlo[i] = mn; thi[i] = mx; }
    } else {
        for (int64_t i = 0; i < total; ++i) {
            int id = (int)((i / stride_dim) % nd);
            tlo[id] = std::min(tlo[id], (double)data[i]);
            thi[id] = std::max(thi[id], (double)data[i]);
        }
    }
    for (int i = 0; i < nd; ++i) { lo[i] += tlo[i]; hi[i] += thi[i]; }
}

template <typename T>
void launch_prep(int D, int grid, const DParams *dp, int a, const int32_t *dsz, int64_t n, int2 *tab) {
    dim3 g(grid), b(256);
    switch (D) {
        case 2: hipLaunchKernelGGL((k_prep_axis_table<T, 2>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
        case 3: hipLaunchKernelGGL((k_prep_axis_table<T, 3>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
        case 4: hipLaunchKernelGGL((k_prep_axis_table<T, 4>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
        case 5: hipLaunchKernelGGL((k_prep_axis_table<T, 5>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
        case 6: hipLaunchKernelGGL((k_prep_axis_table<T, 6>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
        default: break;   // D == 1 has no outer axis
    }
}

template <typename T>
void launch_prep_t(int D, int grid, const DParams *dp, int a, const int32_t *dsz, int64_t n, TabEntry<T> *tab) {
    dim3 g(grid), b(256);
    switch (D) {
        case 1: hipLaunchKernelGGL((k_prep_axis_table_t<T, 1>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
        case 2: hipLaunchKernelGGL((k_prep_axis_table_t<T, 2>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
        case 3: hipLaunchKernelGGL((k_prep_axis_table_t<T, 3>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
        case 4: hipLaunchKernelGGL((k_prep_axis_table_t<T, 4>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
        case 5: hipLaunchKernelGGL((k_prep_axis_table_t<T, 5>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
        default: hipLaunchKernelGGL((k_prep_axis_table_t<T, 6>), g, b, 0, nullptr, dp, a, dsz, n, tab); break;
    }
}

// one outer axis' (cell, t) table over its broadcast domain `dom` (variant 2 / 4), registered for rebuilds
template <typename T>
int build_axis_table(Handle *h, const hjb_problem *p, int a, uint32_t dom, int64_t nent) {
    const int D = p->D, C = p->C;
    const int owned_last = h->hp.n[D - 1];
    DNested::DAxisTable &A = h->hn.at[a];
    std::vector<int32_t> dsz(HJB_MAX_G, 1);
    for (int d = 0; d < D + C; ++d) {
        if (!(dom & (1u << d))) continue;
        dsz[d] = (d < D) ? (d == D - 1 ? owned_last : p->n[d]) : p->m[d - D];
    }
    void *dsz_d = nullptr, *tab = nullptr;
    int st3 = upload(h, dsz, &dsz_d);
    if (st3) return st3;
    st3 = dev_alloc(h, (size_t)nent * sizeof(int2), &tab);
    if (st3) return st3;
    const int grid = (int)std::min<int64_t>((nent + 255) / 256, 65536);
    launch_prep<T>(D, grid, h->dp, a, (const int32_t *)dsz_d, nent, (int2 *)tab);
    h->preps.push_back({a, 0, (const int32_t *)dsz_d, dsz, nent, tab});
    A.tab = tab;
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, sync_setup());
    return HJB_OK;
}

// The axis-0 table of a mode-1 problem that runs without one (axis0_inline): built on demand for the kernels that read
// tables only (forced variant 2, option "axis0_table").
int ensure_axis0_table(Handle *h) {
    if (!h->axis0_inline) return HJB_OK;
    const int st = build_axis_table<float>(h, &h->prob, 0, h->axis0_dom, h->axis0_nent);
    if (st) return st;
    h->axis0_inline = false;
    if (h->packed_pre == 4) h->packed_pre = 1;
    if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }      // captured launches are the other instantiation
    if (h->dn) HIP_TRY(h, hipMemcpy(h->dn, &h->hn, sizeof(DNested), hipMemcpyHostToDevice));
    return HJB_OK;
}

// The halo (planes of the last axis a slab must see beyond the ones it owns) implied by the last axis' next-state terms:
// host arithmetic only, conservative.  Shared by build() and by the partitioners (hjb_create_multi, hjb_rank_create),
// which must not build a whole-grid handle just to learn two integers.
template <typename T>
void halo_from_terms(const hjb_problem *p, bool tab64, int *out_lo, int *out_hi) {
    const int a = p->D - 1, n = p->n[a];
    std::vector<double> lo(n, 0.0), hi(n, 0.0);
    for (int k = 0; k < p->n_next_terms[a]; ++k) {
        if (tab64) term_minmax_along<double>(p, p->next_terms[a][k], a, lo, hi);
        else term_minmax_along<T>(p, p->next_terms[a][k], a, lo, hi);
    }
    std::vector<T> kk(n);
    for (int i = 0; i < n; ++i) kk[i] = (T)p->knots[a][i];
    auto cell_of = [&](double q) {
        int c = (int)(std::upper_bound(kk.begin(), kk.end(), (T)q) - kk.begin()) - 1;
        return std::min(std::max(c, 0), n - 2);
    };
    int need_lo = 0, need_hi = 0;
    for (int i = 0; i < n; ++i) {
        // small relative slack: the sum of per-term extrema is formed in double
        double span = std::fabs(hi[i]) + std::fabs(lo[i]);
        int clo = cell_of(lo[i] - 1e-6 * span), chi = cell_of(hi[i] + 1e-6 * span);
        need_lo = std::max(need_lo, i - clo);
        need_hi = std::max(need_hi, chi + 1 - i);
    }
    *out_lo = need_lo;
    *out_hi = need_hi;
}

template <typename T>
int build(Handle *h, const hjb_problem *p) {
    const int D = p->D, C = p->C;
    DParams &P = h->hp;
    memset(&P, 0, sizeof P);
    P.D = D;
    P.C = C;
    int sb = p->slab_begin, se = p->slab_end, hlo = p->halo_lo, hhi = p->halo_hi;
    if (sb == 0 && se == 0) { se = p->n[D - 1]; hlo = hhi = 0; }
    h->plane0 = sb - hlo;
    h->nplanes = (se + hhi) - h->plane0;
    int64_t s = 1, inner = 1;
    for (int a = 0; a < D; ++a) {
        P.n[a] = (a == D - 1) ? (se - sb) : p->n[a];
        P.jstride[a] = s;
        s *= (a == D - 1) ? h->nplanes : p->n[a];
        if (a < D - 1) inner *= p->n[a];
    }
    h->j_elems = s;
    h->inner = inner;
    h->n_owned = inner * (se - sb);
    h->nU = 1;
    for (int c = 0; c < C; ++c) { P.m[c] = p->m[c]; h->nU *= p->m[c]; }
    for (int c = C; c < HJB_MAX_C; ++c) P.m[c] = 1;
    P.n_owned = h->n_owned;
    P.nU = h->nU;
    P.inner = inner;
    P.plane0 = h->plane0;
    P.nplanes = h->nplanes;
    P.slab_begin = sb;
    P.halo_lo = hlo;
    P.index_base = p->index_base;
    P.idx_bytes = h->idx_bytes;

    const uint32_t state_mask = (1u << D) - 1u;
    for (int a = 0; a < D; ++a) {
        DAxis &ax = P.axis[a];
        const int n = p->n[a];
        std::vector<T> kk(n), rdx(n);
        for (int i = 0; i < n; ++i) kk[i] = (T)p->knots[a][i];
        for (int i = 0; i + 1 < n; ++i) {
            if (!(kk[i + 1] > kk[i]))
                return fail(h, HJB_E_INVALID, "knots of axis %d are not strictly increasing in the working dtype at %d", a, i);
            rdx[i] = (T)1 / (T)(kk[i + 1] - kk[i]);
        }
        rdx[n - 1] = (T)0;
        void *dk = nullptr, *dr = nullptr;
        int st = upload(h, kk, &dk);
        if (st) return st;
        st = upload(h, rdx, &dr);
        if (st) return st;
        ax.knots = dk;
        ax.rdx = dr;
        ax.n = n;
        const double hstep = ((double)kk[n - 1] - (double)kk[0]) / (n - 1);
        double dev = 0;
        for (int i = 0; i < n; ++i) dev = std::max(dev, std::fabs((double)kk[i] - ((double)kk[0] + i * hstep)));
        ax.uniform = dev <= 1.5 * hstep ? 1 : 0;
        ax.x0 = (double)kk[0];
        ax.inv_h = 1.0 / hstep;
        ax.n_terms = p->n_next_terms[a];
        int npre = 0;
        while (npre < ax.n_terms && (p->next_terms[a][npre].mask & ~state_mask) == 0) ++npre;
        ax.n_prefix = npre;
        for (int k = 0; k < ax.n_terms; ++k) {
            // table_dtype F64: the caller's next-state terms are float64.  The float32 copy made here serves the host-side
            // structure analysis only (no stage kernel that evaluates terms is admitted); the tables come from dp64 below
            st = h->tab64 ? make_term<T, double>(h, p, p->next_terms[a][k], &ax.t[k]) : make_term<T>(h, p, p->next_terms[a][k], &ax.t[k]);
            if (st) return st;
        }
    }
    P.n_cost = p->n_cost_terms;
    {
        int npre = 0;
        while (npre < P.n_cost && (p->cost_terms[npre].mask & ~state_mask) == 0) ++npre;
        P.n_cost_prefix = npre;
        P.cost_f64 = h->cost64 ? 1 : 0;
        for (int k = 0; k < P.n_cost; ++k) {
            // cost_dtype F64: the caller's cost terms are float64.  The float32 copy serves the host-side structure analysis
            // only (no stage kernel that sums the cost in float32 is admitted); the kernels read the float64 copy
            int st = h->cost64 ? make_term<T, double>(h, p, p->cost_terms[k], &P.cost[k]) : make_term<T>(h, p, p->cost_terms[k], &P.cost[k]);
            if (!st && h->cost64) st = make_term<double, double>(h, p, p->cost_terms[k], &P.cost64[k]);
            if (st) return st;
        }
    }
    P.model = p->model;
    P.model_h = (float)p->model_h;
    if (p->model == HJB_MODEL_QUAT_EULER321) {
        const size_t ne = (size_t)p->n[0] * p->n[1] * p->n[2];
        for (int i = 0; i < 4; ++i) {
            std::vector<float> v((const float *)p->model_tables[i], (const float *)p->model_tables[i] + ne);
            void *d = nullptr;
            int st = upload(h, v, &d);
            if (st) return st;
            P.model_tab[i] = d;
        }
    }
    // conservative halo implied by the tables of the last axis
    halo_from_terms<T>(p, h->tab64, &h->halo_need_lo, &h->halo_need_hi);
    // ---- variant 1 (control-nested) eligibility --------------------------------
    {
        DNested &N = h->hn;
        memset(&N, 0, sizeof N);
        // division by a launch constant as multiply-high + shifts (Granlund - Montgomery, exact for every 32-bit numerator)
        auto magic = [](int64_t dd, uint32_t *m, int32_t *sh) {
            if (dd <= 1 || dd >= ((int64_t)1 << 31)) { *m = 0; *sh = -1; return; }      // 1: q = r; >= 2^31: the 64-bit-index modes do not use it
            const uint64_t d = (uint64_t)dd;
            int l = 0;
            while (((uint64_t)1 << l) < d) ++l;
            *m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
            *sh = l - 1;
        };
        for (int a = 0; a < D; ++a) magic(P.n[a], &N.div_m[a], &N.div_s[a]);
        magic(P.inner, &N.div_m_inner, &N.div_s_inner);
        const uint32_t in_bit = 1u << (D + C - 1);
        bool ok = !h->tab64;          // variants 1-4 evaluate next-state terms in the kernel, in the problem dtype
        for (int a = 0; a < D - 1 && ok; ++a)
            for (int k = 0; k < p->n_next_terms[a]; ++k)
                if (p->next_terms[a][k].mask & in_bit) ok = false;
        const DAxis &axl = P.axis[D - 1];
        int ax_kin = axl.n_terms, cost_kin = P.n_cost;
        for (int k = axl.n_terms - 1; k >= 0; --k)
            if (p->next_terms[D - 1][k].mask & in_bit) ax_kin = k;
        for (int k = P.n_cost - 1; k >= 0; --k)
            if (p->cost_terms[k].mask & in_bit) cost_kin = k;
        ax_kin = std::max(ax_kin, axl.n_prefix);       // prefix terms are summed per state anyway
        cost_kin = std::max(cost_kin, P.n_cost_prefix);
        N.m_in = p->m[C - 1];
        N.nUo = (int32_t)(h->nU / p->m[C - 1]);
        N.ax_kin = ax_kin;
        N.cost_kin = cost_kin;
        N.n_ax_in = axl.n_terms - ax_kin;
        N.n_cost_in = P.n_cost - cost_kin;
        if (N.n_ax_in > kMaxInAx || N.n_cost_in > kMaxInCost) ok = false;
        int slots = 0;
        for (int s = 0; s < kMaxInner; ++s) { N.in[s].data = nullptr; N.in[s].stride_in = 0; N.in[s].lds_slot = -1; }
        if (ok) {
            for (int s = 0; s < N.n_ax_in; ++s) {
                const DTerm &t = axl.t[ax_kin + s];
                N.in[s].data = t.data;
                N.in[s].stride_in = t.stride[D + C - 1];
                if (p->next_terms[D - 1][ax_kin + s].mask == in_bit) { N.in[s].lds_slot = s; ++slots; }
            }
            for (int s = 0; s < N.n_cost_in; ++s) {
                const DTerm &t = P.cost[cost_kin + s];
                N.in[kMaxInAx + s].data = t.data;
                N.in[kMaxInAx + s].stride_in = t.stride[D + C - 1];
                if (p->cost_terms[cost_kin + s].mask == in_bit) { N.in[kMaxInAx + s].lds_slot = kMaxInAx + s; ++slots; }
            }
        }
        N.n_slots = slots;
        // loop levels (see DNested): o1 runs over control dim C-2, o0 over control dim 0 when C == 3
        N.m_o0 = (C == 3) ? p->m[0] : 1;
        N.m_o1 = (C >= 2) ? p->m[C - 2] : 1;
        magic(N.m_o1, &N.div_m_o1, &N.div_s_o1);
        const uint32_t o1_bit = (C >= 2) ? (1u << (D + C - 2)) : 0u;
        for (int a = 0; a < D; ++a) {
            const DAxis &ax = P.axis[a];
            const int endk = (a == D - 1) ? ax_kin : ax.n_terms;
            int l0 = endk;
            for (int k = endk - 1; k >= ax.n_prefix; --k)
                if (p->next_terms[a][k].mask & o1_bit) l0 = k;
            N.ax_l0[a] = std::max(l0, ax.n_prefix);
        }
        {
            int l0 = cost_kin;
            for (int k = cost_kin - 1; k >= P.n_cost_prefix; --k)
                if (p->cost_terms[k].mask & o1_bit) l0 = k;
            N.cost_l0 = std::max(l0, P.n_cost_prefix);
        }
        h->nested_lds = ((size_t)2 * p->n[D - 1] + (size_t)kMaxInner * (N.m_in + 1)) * sizeof(T);
        h->nested_fast = ok && N.n_ax_in == 1 && N.n_cost_in == 1 && N.in[0].lds_slot >= 0 &&
                         N.in[kMaxInAx].lds_slot >= 0 && ax_kin > 0 && cost_kin > 0;
        if (h->nested_lds > 64 * 1024) ok = false;
        h->nested_ok = ok;
        h->packed_mode = 0;
        // variants 2/4: cost inner term must be a control-only table; the last axis' inner term is either a
        // control-only table b[u_in] (variants 2 and 4) or may also depend on the STATE (variant 4 only:
        // e.g. Solver_attitude.m:425  h*((J1-J2)/J3*X1V.*X2V + U3V/J3)), never on the outer controls
        const uint32_t outer_bits = ((1u << (D + C - 1)) - 1u) & ~((1u << D) - 1u);
        const bool cost_fast = N.n_cost_in == 1 && N.in[kMaxInAx].lds_slot >= 0 && cost_kin > 0;
        const bool ax_gen = N.n_ax_in == 1 && N.in[0].lds_slot < 0 && ax_kin > 0 &&
                            (p->next_terms[D - 1][ax_kin].mask & outer_bits) == 0;
        if (ok && cost_fast && (h->nested_fast || ax_gen) && p->dtype != HJB_F64 && (h->j_elems < ((int64_t)1 << 31) || p->model) &&
            p->n[D - 1] >= 2) {
            bool pk = (ax_kin == P.axis[D - 1].n_prefix) && N.m_in <= kPackedMaxIn;   // last axis: state part + inner term only
            // canonical shape: last axis = state part + b[u_in]; <= 1 cost term per outer loop level;
            // outer axes may have any terms (their cells/weights are precomputed below)
            for (int i = 0; i < HJB_MAX_D + 2; ++i) { memset(&N.ot[i], 0, sizeof N.ot[i]); N.ot[i].lds_off = -1; }
            int32_t ot_floats = 0;
            auto fill = [&](DNested::DOuterTerm &o, const DTerm &t, uint32_t mask, bool first) {
                o.data = t.data;
                for (int a = 0; a < HJB_MAX_D; ++a) o.sstride[a] = a < D ? t.stride[a] : 0;
                o.c0 = (C == 3) ? t.stride[D + 0] : 0;
                o.c1 = (C == 3) ? t.stride[D + 1] : ((C == 2) ? t.stride[D + 0] : 0);
                o.present = 1;
                o.level = (C == 3 && !(mask & (1u << (D + 1)))) ? 0 : 1;
                o.first = first ? 1 : 0;
                o.lds_off = -1;
                o.lds_len = 0;
                if ((mask & ((1u << D) - 1u)) == 0) {      // control-only: stage the whole table in LDS
                    o.lds_len = (int32_t)term_elems(p, mask);
                    o.lds_off = ot_floats;
                    ot_floats += o.lds_len;
                }
            };
            if (pk) {
                const int c0n = N.cost_l0 - P.n_cost_prefix, c1n = cost_kin - N.cost_l0;
                if (c0n > 1 || c1n > 1) pk = false;
                else {
                    if (c0n == 1) {
                        fill(N.ot[HJB_MAX_D], P.cost[P.n_cost_prefix], p->cost_terms[P.n_cost_prefix].mask, P.n_cost_prefix == 0);
                        N.ot[HJB_MAX_D].level = 0;
                    }
                    if (c1n == 1) {
                        fill(N.ot[HJB_MAX_D + 1], P.cost[N.cost_l0], p->cost_terms[N.cost_l0].mask,
                             P.n_cost_prefix == 0 && c0n == 0);
                        N.ot[HJB_MAX_D + 1].level = 1;
                    }
                }
            }
            h->packed_mode = pk ? (h->nested_fast ? 1 : 2) : 0;   // 2: general inner term -> variant 4 only
            h->packed_lds = (size_t)(N.m_in + 1) * 256 * 8 + (size_t)(N.m_in + 1) * 8 + (size_t)2 * p->n[D - 1] * 4 +
                            (size_t)ot_floats * 4;
            {
                const size_t np = (size_t)(N.m_in + 1) / 2;
                h->packed2_lds = (np + 1) * 256 * 8 + (np + 1) * 8 + (size_t)N.m_in * 4 + (size_t)2 * p->n[D - 1] * 4 +
                                 (size_t)ot_floats * 4;
            }
            if (h->packed_lds > 64 * 1024) h->packed_mode = 0;
        }
    }
    void *dst = nullptr;
    int st = dev_alloc(h, sizeof(int32_t), &dst);
    if (st) return st;
    h->d_status = (int32_t *)dst;
    HIP_TRY(h, hipMemset(h->d_status, 0, sizeof(int32_t)));
    P.status = h->d_status;
    void *dpp = nullptr;
    st = dev_alloc(h, sizeof(DParams), &dpp);
    if (st) return st;
    h->dp = (DParams *)dpp;
    HIP_TRY(h, hipMemcpy(h->dp, &P, sizeof(DParams), hipMemcpyHostToDevice));
    if (h->tab64) {
        // float64 shadow of the axes for the table build (k_prep_axis_table_t<double>): knots as given, 1/dx and the
        // next-state terms in double - what griddedInterpolant sees in Solver_pos_att.m:299-327 (double grid vectors,
        // double query tables); the stage kernels never read it
        DParams Q = P;
        for (int a = 0; a < D; ++a) {
            DAxis &ax = Q.axis[a];
            const int n = p->n[a];
            std::vector<double> kk(p->knots[a], p->knots[a] + n), rdx((size_t)n, 0.0);
            for (int i = 0; i + 1 < n; ++i) rdx[(size_t)i] = 1.0 / (kk[(size_t)i + 1] - kk[(size_t)i]);
            void *dk = nullptr, *dr = nullptr;
            int s2 = upload(h, kk, &dk);
            if (!s2) s2 = upload(h, rdx, &dr);
            if (s2) return s2;
            ax.knots = dk;
            ax.rdx = dr;
            const double hstep = (kk[(size_t)n - 1] - kk[0]) / (n - 1);
            double dev = 0;
            for (int i = 0; i < n; ++i) dev = std::max(dev, std::fabs(kk[(size_t)i] - (kk[0] + i * hstep)));
            ax.uniform = dev <= 1.5 * hstep ? 1 : 0;
            ax.x0 = kk[0];
            ax.inv_h = 1.0 / hstep;
            for (int k = 0; k < ax.n_terms; ++k) {
                s2 = make_term<double>(h, p, p->next_terms[a][k], &ax.t[k]);
                if (s2) return s2;
            }
        }
        void *dq = nullptr;
        int s3 = dev_alloc(h, sizeof(DParams), &dq);
        if (s3) return s3;
        h->dp64 = (DParams *)dq;
        HIP_TRY(h, hipMemcpy(h->dp64, &Q, sizeof(DParams), hipMemcpyHostToDevice));
    }
    // ---- variant 2: precompute the stage-invariant (cell, weight) tables of the outer axes -------
    if (h->packed_mode) {
        DNested &N = h->hn;
        const int owned_last = P.n[D - 1];
        size_t total = 0;
        bool fits = true;
        int64_t nent[HJB_MAX_D] = {0};
        uint32_t dom[HJB_MAX_D] = {0};
        for (int a = 0; a < D - 1; ++a) {
            uint32_t m = 0;
            for (int k = 0; k < p->n_next_terms[a]; ++k) m |= p->next_terms[a][k].mask;
            dom[a] = m;
            int64_t ne = 1;
            for (int d = 0; d < D + C; ++d)
                if (m & (1u << d)) ne *= (d < D) ? (d == D - 1 ? owned_last : p->n[d]) : p->m[d - D];
            nent[a] = ne;
            if (ne >= ((int64_t)1 << 31)) fits = false;
            total += (size_t)ne * sizeof(int2);
        }
        if (!fits || total > ((size_t)24 << 30)) {
            h->packed_mode = 0;   // tables too large: variant 1 evaluates on the fly
        } else {
            for (int a = 0; a < D - 1; ++a) {
                DNested::DAxisTable &A = N.at[a];
                memset(&A, 0, sizeof A);
                std::vector<int32_t> dsz(HJB_MAX_G, 1);
                int64_t stride = 1;
                for (int d = 0; d < D + C; ++d) {
                    if (!(dom[a] & (1u << d))) continue;
                    const int sz = (d < D) ? (d == D - 1 ? owned_last : p->n[d]) : p->m[d - D];
                    dsz[d] = sz;
                    if (d < D) A.sstride[d] = (int32_t)stride;
                    else if (C == 3 && d == D + 0) A.c0 = (int32_t)stride;
                    else if ((C == 3 && d == D + 1) || (C == 2 && d == D + 0)) A.c1 = (int32_t)stride;
                    stride *= sz;
                }
                const bool has_o1 = (C >= 2) && (dom[a] & (1u << (D + C - 2)));
                const bool has_o0 = (C == 3) && (dom[a] & (1u << D));
                A.level = has_o1 ? 1 : (has_o0 ? 0 : -1);
                if (p->n_next_terms[a] == 0) continue;     // model axis: evaluated in the stage kernel
                // The C2 shape (mode 1 below: D = 3, three control dims, axis 0 moves with control dim 0, axis 1 with
                // control dim 1): when axis 0's next value is (state-only terms) + ONE term over control dim 0 alone, the
                // stage kernel forms its (cell, t) from q in registers - same ordered sum, same exact search - and the
                // table (8 bytes per state and o0 step: 173 MB on C2, streamed every stage) is not built at all
                if (a == 0 && D == 3 && C == 3 && has_o0 && !has_o1 && h->inline_axis0 &&
                    p->n_next_terms[0] == P.axis[0].n_prefix + 1 && p->next_terms[0][p->n_next_terms[0] - 1].mask == (1u << D)) {
                    uint32_t m1 = 0;
                    for (int k = 0; k < p->n_next_terms[1]; ++k) m1 |= p->next_terms[1][k].mask;
                    if ((m1 & (1u << (D + 1))) && !(m1 & (1u << D))) {       // A.tab stays null
                        h->axis0_inline = true;
                        h->axis0_dom = dom[0];
                        h->axis0_nent = nent[0];
                        continue;
                    }
                }
                void *dsz_d = nullptr, *tab = nullptr;
                int st3 = upload(h, dsz, &dsz_d);
                if (st3) return st3;
                st3 = dev_alloc(h, (size_t)nent[a] * sizeof(int2), &tab);
                if (st3) return st3;
                const int grid = (int)std::min<int64_t>((nent[a] + 255) / 256, 65536);
                launch_prep<T>(D, grid, h->dp, a, (const int32_t *)dsz_d, nent[a], (int2 *)tab);
                h->preps.push_back({a, 0, (const int32_t *)dsz_d, dsz, nent[a], tab});
                A.tab = tab;
            }
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, sync_setup());
            h->packed_pre = 0;
            // modes 1-3 read the level cost terms from LDS only
            const bool cl_lds = (!N.ot[HJB_MAX_D].present || N.ot[HJB_MAX_D].lds_off >= 0) &&
                                (!N.ot[HJB_MAX_D + 1].present || N.ot[HJB_MAX_D + 1].lds_off >= 0);
            if (cl_lds && C == 3 && D == 3 && N.at[0].level == 0 && N.at[1].level == 1) h->packed_pre = 1;
            if (h->axis0_inline && h->packed_pre != 1) {     // mode 1 did not come about after all: the table is needed
                const int st4 = build_axis_table<T>(h, p, 0, dom[0], nent[0]);
                if (st4) return st4;
                h->axis0_inline = false;
            }
            if (h->axis0_inline) h->packed_pre = 4;          // mode 1 without the axis-0 table (kernels_packed2.h MODE 4)
            if (cl_lds && C == 3 && D >= 4 && N.at[D - 3].level == 0 && N.at[D - 2].level == 1) {
                bool pre = true;
                for (int a = 0; a < D - 3; ++a) pre = pre && N.at[a].level < 0;
                if (pre && h->packed2_lds + 36 * 256 * 4 <= 64 * 1024) {
                    h->packed_pre = p->model ? 3 : 2;
                    // Three window planes instead of four (kernels_packed2.h W3P): when the inner control moves the
                    // last axis by less than its narrowest cell per control step, the second cell a sweep enters is a
                    // neighbour of the first.  27 entries and no padding row in the weights: 40 KB per workgroup with 11
                    // torque levels = four workgroups per CU instead of three.  (The kernel still checks every state.)
                    bool near = N.n_ax_in == 1 && p->table_dtype == HJB_TAB_DEFAULT;
                    if (near) {
                        // the last axis' one inner term: (state dims of its mask) x the inner control, control slowest
                        const hjb_term &bt = p->next_terms[D - 1][N.ax_kin];
                        int64_t per_ctrl = 1;
                        for (int d = 0; d < D; ++d)
                            if (bt.mask & (1u << d)) per_ctrl *= p->n[d];
                        const T *bj = (const T *)bt.data;
                        double step = 0.0, width = 1e300;
                        for (int j = 1; j < N.m_in; ++j)
                            for (int64_t e = 0; e < per_ctrl; ++e)
                                step = std::max(step, std::fabs((double)bj[e + j * per_ctrl] - (double)bj[e + (j - 1) * per_ctrl]));
                        for (int i = 1; i < p->n[D - 1]; ++i)
                            width = std::min(width, (double)(T)p->knots[D - 1][i] - (double)(T)p->knots[D - 1][i - 1]);
                        near = step < 0.99 * width;
                    }
                    h->window3_ok = near;
                    {   // visiting order of the 256-state chunks (kernels_packed2.h, option "chunk_order"): when the window slices
                        // of ONE point of the level axes - the whole block of the state-only axes x 27 / 36 entries - outgrow an
                        // XCD's 4 MiB L2, neighbouring chunks of that block must run together (state order); smaller blocks gain
                        // more from the neighbouring points' shared window rows (transposed order).  C3: 51^3 x 27 x 4 B = 14 MB.
                        int64_t blk = 1;
                        for (int a = 0; a + 3 < D; ++a) blk *= p->n[a];
                        h->hn.chunk_order = blk * (int64_t)h->esz * (near ? 27 : 36) > ((int64_t)4 << 20) ? 1 : 0;
                    }
                    if (near) {
                        h->packed_pre += 3;                                        // modes 5 / 6
                        h->packed2_lds += 27 * 256 * 4;
                        h->packed2_lds -= 256 * 8;                                 // no padding row in the weights
                    } else {
                        h->packed2_lds += 36 * 256 * 4;   // the per-state window
                    }
                }
            }
        }
    }
    // ---- variant 5 eligibility: (cell, t) tables of EVERY axis over its own domain (built lazily) ---
    {
        const int owned_last = P.n[D - 1];
        size_t total = 0;
        bool fits = true;
        for (int a = 0; a < D; ++a) {
            uint32_t m = 0;
            for (int k = 0; k < p->n_next_terms[a]; ++k) m |= p->next_terms[a][k].mask;
            h->dom_mask[a] = m;
            int64_t ne = 1;
            for (int d = 0; d < D + C; ++d)
                if (m & (1u << d)) ne *= (d < D) ? (d == D - 1 ? owned_last : p->n[d]) : p->m[d - D];
            h->dom_entries[a] = ne;
            if (ne >= ((int64_t)1 << 31)) fits = false;
            total += (size_t)ne * sizeof(TabEntry<T>);
        }
        // worth it only when the tables are small next to the per-stage work (nS * nU backups)
        const bool small = total <= ((size_t)512 << 20) || (double)total <= 0.5 * (double)h->n_owned * (double)h->nU;
        h->tabled_ok = fits && small && total <= ((size_t)16 << 30);
        {   // the 32-bit form of the table kernel: owned states, the haloed J, every axis table (checked above) and the largest
            // table a cost term can address (global states x controls) all below 2^31 entries
            int64_t ns_global = 1;
            for (int d = 0; d < D; ++d) ns_global *= p->n[d];
            const int64_t lim = kTab32Lim;                                     // (room for the grid-stride step on top of the last index: kernels_tabled.h)
            h->tabled_i32 = h->tabled_ok && h->n_owned < lim && h->j_elems < lim && h->nU < lim &&
                            (double)ns_global * (double)h->nU < (double)lim;
        }
        // variant 6: no axis other than axis 0 may depend on state dim 0 (its cells are then uniform along a row)
        bool rw = h->tabled_ok && D >= 2 && !p->model;
        for (int a = 1; a < D; ++a) rw = rw && (h->dom_mask[a] & 1u) == 0;
        h->row_ok = rw;
        {   // lean form: few controls, 32-bit element offsets, control terms of the cost involve controls only
            bool ln = rw && h->nU <= 64 && h->j_elems * (int64_t)h->esz < ((int64_t)1 << 32) && (P.n_cost - P.n_cost_prefix) <= kLeanMaxCu;
            const uint32_t smask = (1u << D) - 1u;
            for (int k = P.n_cost_prefix; k < P.n_cost; ++k) ln = ln && (p->cost_terms[k].mask & smask) == 0;
            h->row_lean_ok = ln;
        }
        // worth it when rows fill a fair part of the 64-lane waves (C4 120^4: 1.9x over variant 5 in the lean form; 60^4: 1.4x).  Round 4
        // measured the small and odd-sized pos-att grids too (profiles/r04_small_grids.log): the reference's own 30x30x20x15 (47 % of
        // the lanes live, 2.7e5 states) 17.9 against 22.2 us per stage, 33x64x48x32 (52 %) 134 against 170 us, 80x80x60x40 (63 %) 0.52
        // against 0.79 ms - rounds 1 - 3 asked for 70 % and 2^20 states and left those on variant 5
        const double lane_use = (double)p->n[0] / (64.0 * (double)((p->n[0] + 63) / 64));
        // (the lean form only - control terms of the cost over controls alone; with a materialised (state, control) cost table, the
        // mirrors' cost_mode 'exact', the row kernel takes 31 us per stage on that grid against the tabled kernel's 22: the old rule stays)
        h->row_auto = rw && ((h->row_lean_ok && lane_use >= 0.45) || (lane_use >= 0.7 && h->n_owned >= ((int64_t)1 << 20)));
    }
    if (p->model) {
        if (!(h->packed_mode && (h->packed_pre == 3 || h->packed_pre == 6)))
            return fail(h, HJB_E_UNSUPPORTED,
                        "HJB_MODEL_QUAT_EULER321 needs the canonical attitude structure: axis 3 driven by control dim 0, "
                        "axis 4 by control dim 1, axis 5 by control dim 2 (kernels_packed2.h mode 3)");
        h->tabled_ok = false;     // the other stage kernels do not evaluate the model
        h->nested_fast = false;
    }
    if (h->cost64 && !h->tabled_ok)
        return fail(h, HJB_E_UNSUPPORTED, "cost_dtype HJB_COST_F64 is served by the table-driven kernels (variants 5, 7): this grid's per-axis "
                    "(cell, weight) tables do not fit - pass the cost terms in float32 (cost_dtype HJB_COST_DEFAULT)");
    if (h->tab64 && !h->tabled_ok)
        return fail(h, HJB_E_UNSUPPORTED, "table_dtype HJB_TAB_F64 needs the per-axis (cell, weight) tables to fit (variants 5-7): this grid's tables do not - "
                    "pass table_dtype = HJB_TAB_DEFAULT (Python mirrors: table_dtype=None) to run it on float32 queries");
    if (h->nested_ok) {
        void *dnn = nullptr;
        int st2 = dev_alloc(h, sizeof(DNested), &dnn);
        if (st2) return st2;
        h->dn = (DNested *)dnn;
        HIP_TRY(h, hipMemcpy(h->dn, &h->hn, sizeof(DNested), hipMemcpyHostToDevice));
    }
    return HJB_OK;
}

// float64-built entries narrowed to the float32 tables the stage kernels read: the weight is rounded ONCE, here
__global__ void __launch_bounds__(256)
k_tab_narrow(const TabEntry<double> *__restrict__ in, TabEntry<float> *__restrict__ out, int64_t n) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        TabEntry<float> o;
        o.cell = in[e].cell;
        o.t = (float)in[e].t;
        out[e] = o;
    }
}

// One axis' (cell, weight) table in the handle's table dtype.
template <typename T>
int launch_prep_any(Handle *h, int D, int grid, int a, const int32_t *dsz, int64_t n, TabEntry<T> *tab) {
    if constexpr (std::is_same<T, float>::value) {
        if (h->tab64) {
            void *tmp = nullptr;
            // fault injection: hjb_test_hook("fail_tab64_scratch", 1) - an explicit call from inside the process, never the
            // environment - makes this allocation fail (tests/test_gpu_types.py checks that hjb_create then fails instead of
            // handing out a handle on float32 queries)
            if (g_test_fail_tab64_scratch.load() || hipMalloc(&tmp, (size_t)n * sizeof(TabEntry<double>)) != hipSuccess) return fail(h, HJB_E_NOMEM, "float64 table build: scratch of %lld entries", (long long)n);
            launch_prep_t<double>(D, grid, h->dp64, a, dsz, n, (TabEntry<double> *)tmp);
            hipLaunchKernelGGL(k_tab_narrow, dim3(grid), dim3(256), 0, nullptr, (const TabEntry<double> *)tmp, tab, n);
            const hipError_t e1 = sync_setup();
            (void)hipFree(tmp);
            if (e1 != hipSuccess) return fail(h, HJB_E_DEVICE, "float64 table build: %s", hipGetErrorString(e1));
            return HJB_OK;
        }
    }
    launch_prep_t<T>(D, grid, h->dp, a, dsz, n, tab);
    return HJB_OK;
}

template <typename T>
int ensure_tabled_t(Handle *h) {
    if (h->dtb) return HJB_OK;
    const DParams &P = h->hp;
    const int D = P.D, C = P.C;
    const int owned_last = P.n[D - 1];
    DTabled &TBh = h->htb;
    memset(&TBh, 0, sizeof TBh);
    for (int a = 0; a < D; ++a) {
        DTabled::Axis &A = TBh.ax[a];
        std::vector<int32_t> dsz(HJB_MAX_G, 1);
        int64_t stride = 1;
        for (int d = 0; d < D + C; ++d) {
            if (!(h->dom_mask[a] & (1u << d))) continue;
            const int sz = (d < D) ? (d == D - 1 ? owned_last : h->prob.n[d]) : h->prob.m[d - D];
            dsz[d] = sz;
            if (d < D) A.sstride[d] = (int32_t)stride;
            else { A.cstride[d - D] = (int32_t)stride; A.has_ctrl = 1; }
            stride *= sz;
        }
        void *dsz_d = nullptr, *tab = nullptr;
        int st3 = upload(h, dsz, &dsz_d);
        if (st3) return st3;
        if (g_test_fail_tabled_alloc.load()) return fail(h, HJB_E_NOMEM, "(cell, t) table of axis %d: allocation failed (test hook)", a);
        st3 = dev_alloc(h, (size_t)h->dom_entries[a] * sizeof(TabEntry<T>), &tab);
        if (st3) return st3;
        const int grid = (int)std::min<int64_t>((h->dom_entries[a] + 255) / 256, 65536);
        st3 = launch_prep_any<T>(h, D, grid, a, (const int32_t *)dsz_d, h->dom_entries[a], (TabEntry<T> *)tab);
        if (st3) return st3;
        h->preps.push_back({a, 1, (const int32_t *)dsz_d, dsz, h->dom_entries[a], tab});
        A.tab = tab;
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, sync_setup());
    void *d = nullptr;
    int st3 = dev_alloc(h, sizeof(DTabled), &d);
    if (st3) return st3;
    HIP_TRY(h, hipMemcpy(d, &TBh, sizeof(DTabled), hipMemcpyHostToDevice));
    h->dtb = (DTabled *)d;
    return HJB_OK;
}

int ensure_tabled(Handle *h) {
    return h->dtype != HJB_F64 ? ensure_tabled_t<float>(h) : ensure_tabled_t<double>(h);
}

// Rebuild every (cell, weight) table of the handle, with the vector term-sum kernels or - where an axis' last term
// is separable from the others - with the MFMA outer-sum kernel (kernels_prep_mfma.h).  Same bits either way.
static bool prep_split(const Handle *h, const Handle::PrepRec &R, DPrepSplit *S) {
    const hjb_problem &p = h->prob;
    const int a = R.axis, nt = p.n_next_terms[a], G = p.D + p.C;
    if (h->dtype == HJB_F64 || h->tab64 || nt < 2) return false;
    uint32_t others = 0;
    for (int k = 0; k + 1 < nt; ++k) others |= p.next_terms[a][k].mask;
    const uint32_t last = p.next_terms[a][nt - 1].mask;
    if (!last || !others || (last & others)) return false;
    memset(S, 0, sizeof *S);
    int64_t stride = 1, nr = 1, nc = 1;
    for (int d = 0; d < G; ++d) {
        const int sz = R.dsz[(size_t)d];
        const bool in_dom = ((last | others) >> d) & 1u;
        if (!in_dom) { if (sz != 1) return false; continue; }
        if ((last >> d) & 1u) {
            S->col_dim[S->n_col_dims] = d; S->col_size[S->n_col_dims] = sz; S->col_estride[S->n_col_dims++] = (int32_t)stride;
            nc *= sz;
        } else {
            S->row_dim[S->n_row_dims] = d; S->row_size[S->n_row_dims] = sz; S->row_estride[S->n_row_dims++] = (int32_t)stride;
            nr *= sz;
        }
        stride *= sz;
    }
    if (nr * nc != R.n || nr >= ((int64_t)1 << 31) || nc >= ((int64_t)1 << 31)) return false;
    S->n_rows = (int32_t)nr;
    S->n_cols = (int32_t)nc;
    return true;
}

static int rebuild_tables_timed(Handle *h, bool mfma, hipEvent_t e0, hipEvent_t e1);

int rebuild_tables(Handle *h, bool mfma) {
    // the tables are rewritten IN PLACE: wait for the whole device, not only for the null stream - a stage may still be in flight
    // on the handle's own stream, a rank's strip streams or a caller's stream (hjb_backup_stage_device).  Not a hot path.  
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(h, HJB_E_DEVICE, "hipEventCreate failed"); }
    const int st = rebuild_tables_timed(h, mfma, e0, e1);
    (void)hipEventDestroy(e0);         // one exit: the events never leak
    (void)hipEventDestroy(e1);
    return st;
}

static int rebuild_tables_timed(Handle *h, bool mfma, hipEvent_t e0, hipEvent_t e1) {
    HIP_TRY(h, hipEventRecord(e0, nullptr));
    const int D = h->hp.D;
    int n_mfma = 0;
    for (const auto &R : h->preps) {
        DPrepSplit S;
        if (mfma && prep_split(h, R, &S)) {
            const int64_t tiles = (int64_t)((S.n_rows + 31) / 32) * ((S.n_cols + 31) / 32);
            dim3 g((unsigned)std::min<int64_t>((tiles + 3) / 4, 65536)), b(256);
            switch (D) {
                case 1: hipLaunchKernelGGL((k_prep_axis_table_mfma<1>), g, b, 0, nullptr, h->dp, R.axis, S, (int2 *)R.tab); break;
                case 2: hipLaunchKernelGGL((k_prep_axis_table_mfma<2>), g, b, 0, nullptr, h->dp, R.axis, S, (int2 *)R.tab); break;
                case 3: hipLaunchKernelGGL((k_prep_axis_table_mfma<3>), g, b, 0, nullptr, h->dp, R.axis, S, (int2 *)R.tab); break;
                case 4: hipLaunchKernelGGL((k_prep_axis_table_mfma<4>), g, b, 0, nullptr, h->dp, R.axis, S, (int2 *)R.tab); break;
                case 5: hipLaunchKernelGGL((k_prep_axis_table_mfma<5>), g, b, 0, nullptr, h->dp, R.axis, S, (int2 *)R.tab); break;
                default: hipLaunchKernelGGL((k_prep_axis_table_mfma<6>), g, b, 0, nullptr, h->dp, R.axis, S, (int2 *)R.tab); break;
            }
            ++n_mfma;
            continue;
        }
        const int grid = (int)std::min<int64_t>((R.n + 255) / 256, 65536);
        if (h->dtype == HJB_F64) {
            if (R.kind == 0) launch_prep<double>(D, grid, h->dp, R.axis, R.dsz_d, R.n, (int2 *)R.tab);
            else launch_prep_t<double>(D, grid, h->dp, R.axis, R.dsz_d, R.n, (TabEntry<double> *)R.tab);
        } else {
            if (R.kind == 0) launch_prep<float>(D, grid, h->dp, R.axis, R.dsz_d, R.n, (int2 *)R.tab);
            else { const int pst = launch_prep_any<float>(h, D, grid, R.axis, R.dsz_d, R.n, (TabEntry<float> *)R.tab); if (pst) return pst; }
        }
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(e1, nullptr));
    HIP_TRY(h, hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
    h->prep_us = (double)ms * 1e3;
    h->prep_mfma = mfma ? 1 : 0;
    h->prep_mfma_axes = n_mfma;
    return HJB_OK;
}

int table_hash(Handle *h, uint64_t *out) {      // FNV-1a over the bytes of every table, in registration order
    uint64_t hsh = 1469598103934665603ull;
    std::vector<unsigned char> buf;
    for (const auto &R : h->preps) {
        const size_t bytes = (size_t)R.n * ((h->dtype == HJB_F64 && R.kind == 1) ? 16 : 8);
        buf.resize(bytes);
        HIP_TRY(h, hipMemcpy(buf.data(), R.tab, bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < bytes; ++i) { hsh ^= buf[i]; hsh *= 1099511628211ull; }
    }
    *out = hsh;
    return HJB_OK;
}

// ---- variant 7 (kernels_colsweep.h): eligibility + the per-(i2, i3) plan, built once on the host from the
// variant-5 tables of axes 2 and 3 (tiny: n2 * n3 * nU entries) --------------------------------------------------
template <typename T>
bool colsweep_plan(Handle *h, int gax, const std::vector<TabEntry<T>> (&tab)[2], const std::vector<std::vector<T>> &cu,
                   std::vector<int32_t> &plan, int64_t *rows_total, int *ng_max, std::vector<int32_t> &cells) {
    static_assert(sizeof(T) == 4, "plan words are 32-bit");
    const DParams &P = h->hp;
    const int n2 = P.n[2], n3 = P.n[3], nU = (int)h->nU, wax = 5 - gax;
    const int64_t gs = P.jstride[gax], ws = P.jstride[wax];
    const int nwk = wax == 3 ? h->nplanes : P.n[wax];        // knots of the window axis present in this handle's J buffers
    if (nwk < 3) return false;
    plan.assign((size_t)n2 * n3 * kCsPlanWords, 0);
    cells.assign((size_t)n2 * n3 * kCsGMax * 2, 0);         // (group-axis cell, first window knot) of every group
    *rows_total = 0;
    *ng_max = 1;
    int mid_rows = 0;
    auto bits = [](T x) { int32_t b; memcpy(&b, &x, 4); return b; };
    for (int i3 = 0; i3 < n3; ++i3) {
        for (int i2 = 0; i2 < n2; ++i2) {
            int32_t *q = &plan[(size_t)(i2 + n2 * i3) * kCsPlanWords];
            // a group: the cell cg of the group axis, window knots wmin .. wmin + 2 of the other axis, member slots
            // [0, MMAX/2) (window cell wmin) and [MMAX/2, MMAX) (window cell wmin + 1)
            struct Grp { int cg, wmin, slot[kCsMMax]; };
            Grp grp[kCsGMax];
            int ng = 0, bad = 0;
            int cc[2][kCsUMax];
            T tt[2][kCsUMax];
            for (int u = 0; u < nU; ++u) {
                for (int a = 2; a < 4; ++a) {
                    const DTabled::Axis &A = h->htb.ax[a];
                    const TabEntry<T> &e = tab[a - 2][(size_t)(A.sstride[2] * i2 + A.sstride[3] * i3 + A.cstride[0] * u)];
                    int c = e.cell;
                    if (a == 3) {                       // global plane -> plane of this handle's J buffers
                        c -= h->plane0;
                        if (c < 0 || c + 1 >= h->nplanes) { bad = 1; c = c < 0 ? 0 : h->nplanes - 2; }
                    }
                    cc[a - 2][u] = c;
                    tt[a - 2][u] = e.t;
                }
            }
            // windows per group-axis cell: the smallest uncovered window cell opens a window of two cells
            for (int u = 0; u < nU; ++u) {
                const int cg = cc[gax - 2][u], cw = cc[wax - 2][u];
                int wmin = cw;                              // the window this control belongs to: greedy cover, walked
                {                                           // from the smallest window cell among the controls of cg
                    int start = cw;
                    for (int v = 0; v < nU; ++v) if (cc[gax - 2][v] == cg) start = std::min(start, cc[wax - 2][v]);
                    for (;;) {
                        if (cw <= start + 1) { wmin = start; break; }
                        int nxt = cw;                       // the next uncovered cell opens the next window
                        for (int v = 0; v < nU; ++v)
                            if (cc[gax - 2][v] == cg && cc[wax - 2][v] > start + 1) nxt = std::min(nxt, cc[wax - 2][v]);
                        start = nxt;
                    }
                }
                // three knots wmin .. wmin + 2 must exist: the last window of the axis starts one knot lower
                if (wmin + 2 > nwk - 1) wmin = nwk - 3;
                const int pair = cw - wmin;                 // 0 or 1
                constexpr int PS = kCsMMax / 2;             // slots per window pair
                auto free_slot = [&](const Grp &G) {
                    for (int s = pair * PS; s < (pair + 1) * PS; ++s) if (G.slot[s] < 0) return s;
                    return -1;
                };
                int g = 0;
                for (; g < ng; ++g)
                    if (grp[g].cg == cg && grp[g].wmin == wmin && free_slot(grp[g]) >= 0) break;
                if (g == ng) {
                    if (ng == kCsGMax) return false;
                    grp[ng].cg = cg; grp[ng].wmin = wmin;
                    for (int s = 0; s < kCsMMax; ++s) grp[ng].slot[s] = -1;
                    ++ng;
                }
                grp[g].slot[free_slot(grp[g])] = u;
            }
            *ng_max = std::max(*ng_max, ng);
            q[0] = bad | (ng << 8);
            // visit the groups in ascending order of their highest control: fewer slots then come after a higher-numbered
            // control and need the (value, control number) comparison
            auto gmax = [&](const Grp &G) { int mx = -1; for (int s = 0; s < kCsMMax; ++s) mx = std::max(mx, G.slot[s]); return mx; };
            std::stable_sort(grp, grp + ng, [&](const Grp &a, const Grp &b) { return gmax(a) < gmax(b); });
            for (int g = 0; g < ng; ++g) {              // ascending control numbers inside each window pair
                std::sort(grp[g].slot, grp[g].slot + kCsMMax / 2, [](int a, int b) { return (unsigned)a < (unsigned)b; });
                std::sort(grp[g].slot + kCsMMax / 2, grp[g].slot + kCsMMax, [](int a, int b) { return (unsigned)a < (unsigned)b; });
            }
            int seen_max = -1;
            for (int g = 0; g < kCsGMax; ++g) {
                const Grp &G = grp[g < ng ? g : 0];                 // padding: a member-less copy of group 0's rows
                const int64_t off = (gs * G.cg + ws * G.wmin) * (int64_t)h->esz;
                const int nw = 3;
                int usedbits = 0;
                q[1 + g] = (int32_t)(uint32_t)off;
                cells[((size_t)(i2 + n2 * i3) * kCsGMax + g) * 2] = G.cg;
                cells[((size_t)(i2 + n2 * i3) * kCsGMax + g) * 2 + 1] = G.wmin;
                if (g < ng) {
                    // the kernels stop at a pair's first empty slot: used slots are a prefix of each pair
                    for (int pr = 0; pr < 2; ++pr)
                        for (int sidx = pr * (kCsMMax / 2) + 1; sidx < (pr + 1) * (kCsMMax / 2); ++sidx)
                            if (G.slot[sidx] >= 0 && G.slot[sidx - 1] < 0) return false;
                    *rows_total += 2 * nw;
                    for (int sidx = 0; sidx < kCsMMax; ++sidx) {
                        const int u = G.slot[sidx];
                        if (u < 0) continue;
                        usedbits |= 1 << sidx;
                        if (u < seen_max) usedbits |= 0x10000 << sidx;
                        seen_max = std::max(seen_max, u);
                        int32_t *sl = q + kCsPI + 8 * (g * kCsMMax + sidx);
                        sl[0] = bits(tt[wax - 2][u]);
                        sl[1] = bits(tt[gax - 2][u]);
                        sl[3] = u;
                        for (size_t k = 0; k < cu.size(); ++k) sl[k == 0 ? 2 : 3 + k] = bits(cu[k][(size_t)u]);
                        if (!h->cs_cu64.empty()) memcpy(&sl[4], &h->cs_cu64[(size_t)u], sizeof(double));
                    }
                }
                q[1 + kCsGMax + g] = usedbits | (nw << 8);
                if (i2 == n2 / 2 && i3 == n3 / 2 && g < ng)
                    mid_rows += 2 * (1 + ((usedbits & 7) != 0) + ((usedbits & 0x38) != 0));
            }
        }
    }
    h->cs_rows_mid = mid_rows;
    return true;
}

// Column -> XCD assignment of variant 7 (DColSweep::xcd_ig): group-axis indices sorted by (index mod M, index), cut
// into 8 equal parts.  Default M = 1: plain contiguous ranges; option "cs_xcd_mod" sets M, -1 = the spacing of the
// groups' cells in a mid-grid plan.
int colsweep_map(Handle *h, const std::vector<int32_t> &plan) {
    const DParams &P = h->hp;
    DColSweep &CSh = h->hcs;
    const int gax = CSh.gax, n2 = P.n[2], n3 = P.n[3];
    CSh.xcd_win = h->cs_xcd_axis ? 1 : 0;
    const int ngx = CSh.xcd_win ? P.n[5 - gax] : P.n[gax];          // indices of the axis the XCDs split
    int M = CSh.xcd_win ? 1 : h->cs_xcd_mod;
    if (M == 0) M = 1;           // measured on C4 (120^4 x 9): contiguous ranges 2.67 ms per stage, residue classes of the
                                 // group spacing (cs_xcd_mod = -1) 2.84 ms
    if (M < 0) {
        // spacing of the distinct group cells of the middle column, from the row offsets of its plan
        const int32_t *q = &plan[(size_t)(n2 / 2 + n2 * (n3 / 2)) * kCsPlanWords];
        const int ng = q[0] >> 8;
        const int64_t gb = P.jstride[gax] * (int64_t)h->esz, wb = P.jstride[5 - gax] * (int64_t)h->esz;
        std::vector<int64_t> cells;
        for (int g = 0; g < ng; ++g) {
            // row offset = gs * cg + ws * wmin (bytes): the group-axis cell is the quotient by the larger stride
            const int64_t off = (uint32_t)q[1 + g];
            cells.push_back(gax == 3 ? off / gb : (off % wb) / gb);
        }
        std::sort(cells.begin(), cells.end());
        cells.erase(std::unique(cells.begin(), cells.end()), cells.end());
        int64_t best = 0;
        for (size_t i = 1; i < cells.size(); ++i) best = best == 0 ? cells[i] - cells[i - 1] : std::min(best, cells[i] - cells[i - 1]);
        M = (int)std::max<int64_t>(1, std::min<int64_t>(best, ngx));
    }
    std::vector<int> order((size_t)ngx);
    for (int i = 0; i < ngx; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (a % M) < (b % M); });
    const int stride = (ngx + 7) / 8;
    std::vector<int32_t> tab((size_t)8 * stride, 0);
    for (int x = 0; x < 8; ++x) {
        const int b = (int)((int64_t)ngx * x / 8), e = (int)((int64_t)ngx * (x + 1) / 8);
        CSh.xcd_cnt[x] = e - b;
        for (int i = b; i < e; ++i) tab[(size_t)x * stride + (i - b)] = order[(size_t)i];
    }
    CSh.xcd_stride = stride;
    void *d = nullptr;
    const int st = upload(h, tab, &d);
    if (st) return st;
    CSh.xcd_ig = (const int32_t *)d;
    return HJB_OK;
}

// One-load form of variant 7: in every wave of kCsDppLanes consecutive axis-0 states, (cell - state index) is the same
// for all states but at most one (a cell clamped at the grid edge; the kernel gives that state a lane pair of its own).
template <typename T>
int colsweep_dpp_ok(Handle *h, bool *ok) {
    const DParams &P = h->hp;
    const DTabled::Axis &A0 = h->htb.ax[0];
    std::vector<TabEntry<T>> tab((size_t)h->dom_entries[0]);
    HIP_TRY(h, hipMemcpy(tab.data(), A0.tab, tab.size() * sizeof(TabEntry<T>), hipMemcpyDeviceToHost));
    const int n0 = P.n[0];
    const int r2 = A0.sstride[2] ? P.n[2] : 1, r3 = A0.sstride[3] ? P.n[3] : 1;
    *ok = true;
    for (int i3 = 0; i3 < r3 && *ok; ++i3)
        for (int i2 = 0; i2 < r2 && *ok; ++i2)
            for (int c = 0; c < n0 && *ok; c += kCsDppLanes) {
                const int e = std::min(n0, c + kCsDppLanes);
                auto rel = [&](int i0) { return tab[(size_t)(A0.sstride[0] * i0 + A0.sstride[2] * i2 + A0.sstride[3] * i3)].cell - i0; };
                // the common value is the one at least half of the states take (ties: the first state's, as in the kernel)
                const int r0 = rel(c);
                int same = 0;
                for (int i0 = c; i0 < e; ++i0) same += rel(i0) == r0;
                int kb = r0;
                if (2 * same < e - c)
                    for (int i0 = c; i0 < e; ++i0) if (rel(i0) != r0) { kb = rel(i0); break; }
                int odd = 0;
                for (int i0 = c; i0 < e; ++i0) odd += rel(i0) != kb;
                if (odd > 1) *ok = false;
            }
    return HJB_OK;
}

// Cooperative form of variant 7 (kernels_colcoop.h): a workgroup = kCcW columns that are neighbours along the window
// axis.  It applies when axis 1's cell does not depend on the window-axis index (the workgroup steps through one
// sequence of axis-1 knots), every workgroup's columns need at most `ng` distinct group-axis cells with window knots
// inside kCcNV staged ones, and their axis-0 cells inside kCcXW staged knots.  Fills plan word [1 + 2 GMAX + g] (the
// group's first corner row among the staged rows) and the per-workgroup words; sets h->cs_coop_epl.
template <typename T>
int colcoop_plan(Handle *h, std::vector<int32_t> &plan, const std::vector<int32_t> &cells, std::vector<int32_t> &wgw) {
    h->cs_coop_epl = 0;
    const DParams &P = h->hp;
    const DColSweep &CSh = h->hcs;
    const int gax = CSh.gax, wax = 5 - gax, n0 = P.n[0], n2 = P.n[2], n3 = P.n[3];
    h->cs_coop_why = 1;
    if (CSh.ng > kCcNCG) return HJB_OK;
    h->cs_coop_why = 2;
    if (h->dom_mask[1] & (1u << wax)) return HJB_OK;
    h->cs_coop_why = 3;
    const int epl = h->esz == 4 ? (n0 % 4 == 0 ? 4 : 0) : (h->esz == 2 ? (n0 % 8 == 0 ? 8 : 0) : 0);
    if (!epl || n0 < epl) return HJB_OK;
    const int nwk = wax == 3 ? h->nplanes : P.n[wax];
    const int ngx = P.n[gax], nwax = P.n[wax];
    const int chunks = (n0 + 63) / 64, nblk = (nwax + kCcW - 1) / kCcW;
    const int xw = h->esz == 2 ? kCcXWh : kCcXW;
    const int rowb = xw * (int)h->esz;
    const DTabled::Axis &A0 = h->htb.ax[0];
    std::vector<TabEntry<T>> tab0((size_t)h->dom_entries[0]);
    HIP_TRY(h, hipMemcpy(tab0.data(), A0.tab, tab0.size() * sizeof(TabEntry<T>), hipMemcpyDeviceToHost));
    wgw.assign((size_t)ngx * chunks * nblk * kCcWgWords, 0);
    auto col = [&](int ig, int iw) { return gax == 3 ? (size_t)(iw + n2 * ig) : (size_t)(ig + n2 * iw); };
    for (int ig = 0; ig < ngx; ++ig)
        for (int blk = 0; blk < nblk; ++blk) {
            // distinct group-axis cells of the block's columns, the window knots each needs
            int cg[kCcNCG], vmin[kCcNCG], vmax[kCcNCG], ncg = 0;
            for (int j = 0; j < kCcW; ++j) {
                const int iw = std::min(blk * kCcW + j, nwax - 1);
                const size_t c = col(ig, iw);
                const int ng = plan[c * kCsPlanWords] >> 8;
                for (int g = 0; g < ng; ++g) {
                    const int cgv = cells[(c * kCsGMax + g) * 2], wm = cells[(c * kCsGMax + g) * 2 + 1];
                    int ci = 0;
                    while (ci < ncg && cg[ci] != cgv) ++ci;
                    if (ci == ncg) {
                        if (ncg == CSh.ng) { h->cs_coop_why = 4; return HJB_OK; }
                        cg[ncg] = cgv; vmin[ncg] = wm; vmax[ncg] = wm + 2; ++ncg;
                    } else {
                        vmin[ci] = std::min(vmin[ci], wm);
                        vmax[ci] = std::max(vmax[ci], wm + 2);
                    }
                }
            }
            for (int ci = 0; ci < ncg; ++ci)
                if (vmax[ci] - vmin[ci] + 1 > kCcNV) { h->cs_coop_why = 5; return HJB_OK; }
            for (int j = 0; j < kCcW; ++j) {
                const int iw = blk * kCcW + j;
                if (iw >= nwax) break;
                const size_t c = col(ig, iw);
                for (int g = 0; g < kCsGMax; ++g) {         // padded groups repeat group 0's rows, like their global offsets
                    const int cgv = cells[(c * kCsGMax + g) * 2], wm = cells[(c * kCsGMax + g) * 2 + 1];
                    int ci = 0;
                    while (ci < ncg && cg[ci] != cgv) ++ci;
                    plan[c * kCsPlanWords + 1 + 2 * kCsGMax + g] = ((ci * 2) * kCcNV + (wm - vmin[ci])) * rowb;
                }
            }
            for (int chunk = 0; chunk < chunks; ++chunk) {
                int32_t *q = &wgw[((size_t)(ig * chunks + chunk) * nblk + blk) * kCcWgWords];
                int cmin = INT32_MAX, cmax = INT32_MIN;
                for (int j = 0; j < kCcW; ++j) {
                    const int iw = std::min(blk * kCcW + j, nwax - 1);
                    const int i2 = gax == 3 ? iw : ig, i3 = gax == 3 ? ig : iw;
                    for (int i0 = chunk * 64; i0 < std::min(n0, chunk * 64 + 64); ++i0) {
                        const int c0 = tab0[(size_t)(A0.sstride[0] * i0 + A0.sstride[2] * i2 + A0.sstride[3] * i3)].cell;
                        cmin = std::min(cmin, c0);
                        cmax = std::max(cmax, c0);
                    }
                }
                const int xlo = cmin / epl * epl;
                if (cmin < 0 || cmax + 1 - xlo > xw - 1) { h->cs_coop_why = 6; return HJB_OK; }
                q[0] = xlo;
                q[1] = ncg;
                for (int ci = 0; ci < ncg; ++ci) {
                    q[2 + ci] = (int32_t)(uint32_t)((P.jstride[gax] * (int64_t)cg[ci] + P.jstride[wax] * (int64_t)vmin[ci]) * (int64_t)h->esz);
                    q[2 + kCcNCG + ci] = std::min(kCcNV, nwk - vmin[ci]);
                }
            }
        }
    h->cs_coop_why = 0;
    h->cs_coop_epl = epl;
    return HJB_OK;
}

// Variant 7: in how many parts (waves) a column is swept (DColSweep::split).  Automatic: doubled while the launch stays
// within five times the chip's 6144 wave slots (6 waves per SIMD) and every part keeps >= 12 steps (a part starts by
// priming: about a step and a half of extra gathers).  Measured on one middle rank of an 8-GPU run of C4 (15 planes =
// 3600 columns, profiles/r02_rank_slab_timing.log): 1 / 2 / 4 / 8 parts -> 0.270 / 0.249 / 0.233 / 0.235 ms per stage;
// a boundary strip (240 columns) lasts 15 steps instead of 120.  Round 4, whole grids (launches far beyond the wave slots): parts of
// ~60 steps beat one long column by 1-2 % on every shape tried (120^4: 1 / 2 / 3 parts 1.674 / 1.640 / 1.647 ms; 160 steps: 1.551 /
// 1.527 / 1.516; 80 steps: equal; profiles/r04_c4_split.log) - so a column is also cut into round(n1 / 60) parts.
// The device copy of the column-sweep parameters, with the launch record at its head (kernels_colsweep.h CsRec): every
// scalar a wave reads before it knows its column, copied from the structures that own them.
int colsweep_upload(Handle *h) {
    DColSweep &C = h->hcs;
    const DParams &P = h->hp;
    const DTabled &T = h->htb;
    uint32_t *r = C.rec;
    memset(r, 0, sizeof C.rec);
    auto put_ptr = [&](int i, const void *p) { const uint64_t v = (uint64_t)(uintptr_t)p; r[i] = (uint32_t)v; r[i + 1] = (uint32_t)(v >> 32); };
    for (int x = 0; x < 8; ++x) r[kRecXcdCnt + x] = (uint32_t)C.xcd_cnt[x];
    r[kRecN0] = (uint32_t)P.n[0]; r[kRecN1] = (uint32_t)P.n[1]; r[kRecN2] = (uint32_t)P.n[2]; r[kRecN3] = (uint32_t)P.n[3];
    r[kRecSplit] = (uint32_t)C.split; r[kRecWin] = (uint32_t)C.xcd_win; r[kRecXStride] = (uint32_t)C.xcd_stride; r[kRecNcu] = (uint32_t)C.ncu;
    put_ptr(kRecXcdIg, C.xcd_ig); put_ptr(kRecPlan, C.plan);
    put_ptr(kRecA0Tab, T.ax[0].tab); put_ptr(kRecA1Tab, T.ax[1].tab); put_ptr(kRecStatus, P.status);
    r[kRecGBytes] = C.g_bytes; r[kRecWBytes] = C.w_bytes; r[kRecS1Bytes] = C.s1_bytes;
    r[kRecNpreCol] = (uint32_t)C.npre_col; r[kRecNpre] = (uint32_t)P.n_cost_prefix; r[kRecStepUniform] = (uint32_t)C.step_uniform;
    r[kRecA0S0] = (uint32_t)T.ax[0].sstride[0]; r[kRecA0S2] = (uint32_t)T.ax[0].sstride[2]; r[kRecA0S3] = (uint32_t)T.ax[0].sstride[3];
    r[kRecA1S1] = (uint32_t)T.ax[1].sstride[1]; r[kRecA1S2] = (uint32_t)T.ax[1].sstride[2]; r[kRecA1S3] = (uint32_t)T.ax[1].sstride[3];
    r[kRecSlabBegin] = (uint32_t)P.slab_begin; r[kRecHaloLo] = (uint32_t)P.halo_lo;
    r[kRecJs1] = (uint32_t)P.jstride[1]; r[kRecJs2] = (uint32_t)P.jstride[2]; r[kRecJs3] = (uint32_t)P.jstride[3];
    r[kRecIndexBase] = (uint32_t)P.index_base; r[kRecIdxBytes] = (uint32_t)P.idx_bytes;
    // the cost record: up to three column-constant state terms (those before the first that depends on state dim 1) and the one
    // per-step term of the usual shape; a shape it cannot hold says so (n = -1 never equals npre_col) and the kernel reads DParams
    uint32_t *c = C.crec;
    memset(c, 0, sizeof C.crec);
    const bool holds = !h->cost64 && C.npre_col >= 0 && C.npre_col <= 3;
    c[kCRecNCol] = holds ? (uint32_t)C.npre_col : (uint32_t)-1;
    if (holds) {
        auto put_term = [&](int at, const DTerm &t, int sa, int sb, int sc) {
            const uint64_t v = (uint64_t)(uintptr_t)t.data;
            c[at] = (uint32_t)v; c[at + 1] = (uint32_t)(v >> 32);
            c[at + 2] = (uint32_t)t.stride[sa]; c[at + 3] = (uint32_t)t.stride[sb]; c[at + 4] = (uint32_t)t.stride[sc];
        };
        for (int j = 0; j < C.npre_col; ++j) put_term(kCRecTerm + 5 * j, P.cost[j], 0, 2, 3);
        if (C.step_uniform && P.n_cost_prefix - C.npre_col == 1) {
            c[kCRecHasSu] = 1;
            put_term(kCRecSu, P.cost[C.npre_col], 1, 2, 3);
        }
    }
    if (!h->dcs) return fail(h, HJB_E_DEVICE, "variant 7 parameters not allocated");
    HIP_TRY(h, hipMemcpy(h->dcs, &C, sizeof(DColSweep), hipMemcpyHostToDevice));
    return HJB_OK;
}

void colsweep_split(Handle *h) {
    const DParams &P = h->hp;
    DColSweep &CSh = h->hcs;
    const int lanes = CSh.dpp ? kCsDppLanes : 64;
    const int64_t chunks = (P.n[0] + lanes - 1) / lanes;
    const int64_t waves = chunks * (int64_t)P.n[2] * (int64_t)P.n[3];
    const int n1 = P.n[1];
    int S = h->cs_split;
    if (S <= 0) {
        S = 1;
        // (five rounds of the 6144 wave slots at six waves per SIMD; rounds 2 - 3 said three rounds of 5120: a middle rank of a 4-GPU run of
        // C4 - 7200 columns - in 2 / 3 / 4 parts 0.430 / 0.419 / 0.416 ms fused, 0.456 / 0.440 / 0.438 with its strips beside the interior)
        while (S < 8 && waves * S * 2 <= 5 * 6144 && n1 / (S * 2) >= 12) S *= 2;
        // launches below one round of the wave slots (the reference's own 30x30x20x15 grid: 450 columns of 20 steps): parts as short as
        // five steps still pay - 31.3 / 18.7 / 12.8 us per stage in 1 / 2 / 4 parts (profiles/r04_small_grids.log)
        while (S < 8 && waves * S * 2 <= 4096 && n1 / (S * 2) >= 5) S *= 2;
        // round 5: with the one-round-trip prime and the batched set-up a part costs little to start, and such a launch is ONE wave's
        // critical path (5.6 us + 1.14 us per step on that grid): as many parts as fit three quarters of the wave slots, two steps
        // each at least - 17.0 / 11.3 / 10.6 -> 10.0 us per stage in 2 / 4 / 10 parts (profiles/r05_small_grids.log)
        if (waves * S <= 4608 && n1 >= 4 && n1 <= 40) S = (int)std::max<int64_t>(S, std::min<int64_t>(std::min<int64_t>(n1 / 2, 4608 / std::max<int64_t>(waves, 1)), 16));
        S = std::max(S, std::min(8, (n1 + 30) / 60));
    }
    CSh.split = std::max(1, std::min(S, std::max(1, n1)));
}

template <typename T>
int ensure_colsweep_t(Handle *h) {
    if (h->cs_state >= 0) return HJB_OK;
    h->cs_state = 0;
    const DParams &P = h->hp;
    if (P.D != 4 || P.C != 1 || P.model || !h->tabled_ok || h->nU > kCsUMax) return HJB_OK;
    if (h->j_elems * (int64_t)h->esz >= ((int64_t)1 << 32) || h->n_owned >= ((int64_t)1 << 31)) return HJB_OK;
    const uint32_t cbit = 1u << 4;
    if ((h->dom_mask[0] & (cbit | 2u)) || (h->dom_mask[1] & (cbit | 1u)) || (h->dom_mask[2] & 3u) || (h->dom_mask[3] & 3u)) return HJB_OK;
    const int ncu = P.n_cost - P.n_cost_prefix;
    if (ncu > kCsMaxCu) return HJB_OK;
    for (int k = P.n_cost_prefix; k < P.n_cost; ++k)
        if (h->prob.cost_terms[k].mask != cbit) return HJB_OK;
    int npre_col = 0;
    while (npre_col < P.n_cost_prefix && (h->prob.cost_terms[npre_col].mask & 2u) == 0) ++npre_col;
    bool step_uniform = true;
    for (int k = npre_col; k < P.n_cost_prefix; ++k) step_uniform = step_uniform && (h->prob.cost_terms[k].mask & 1u) == 0;
    int st = ensure_tabled(h);
    if (st) return st;
    std::vector<TabEntry<T>> tab[2];
    for (int a = 2; a < 4; ++a) {
        tab[a - 2].resize((size_t)h->dom_entries[a]);
        HIP_TRY(h, hipMemcpy(tab[a - 2].data(), h->htb.ax[a].tab, tab[a - 2].size() * sizeof(TabEntry<T>), hipMemcpyDeviceToHost));
    }
    std::vector<std::vector<T>> cu((size_t)ncu, std::vector<T>((size_t)h->nU));
    for (int k = 0; k < ncu; ++k)
        HIP_TRY(h, hipMemcpy(cu[(size_t)k].data(), P.cost[P.n_cost_prefix + k].data, (size_t)h->nU * sizeof(T), hipMemcpyDeviceToHost));
    h->cs_cu64.clear();
    if (h->cost64 && ncu == 1) {       // the one control term in float64: a slot carries it in words 4, 5 (cost form 2)
        h->cs_cu64.resize((size_t)h->nU);
        HIP_TRY(h, hipMemcpy(h->cs_cu64.data(), P.cost64[P.n_cost_prefix].data, (size_t)h->nU * sizeof(double), hipMemcpyDeviceToHost));
    }
    // group by the axis that leaves fewer corner rows to load
    std::vector<int32_t> plan[2];
    int64_t rows[2] = {0, 0};
    int ngm[2] = {1, 1};
    std::vector<int32_t> cells[2];
    const bool ok3 = colsweep_plan<T>(h, 3, tab, cu, plan[1], &rows[1], &ngm[1], cells[1]);
    const bool ok2 = colsweep_plan<T>(h, 2, tab, cu, plan[0], &rows[0], &ngm[0], cells[0]);
    if (!ok2 && !ok3) return HJB_OK;
    const int pick = (ok3 && (!ok2 || ngm[1] < ngm[0] || (ngm[1] == ngm[0] && rows[1] <= rows[0]))) ? 1 : 0;
    DColSweep &CSh = h->hcs;
    memset(&CSh, 0, sizeof CSh);
    CSh.gax = pick ? 3 : 2;
    CSh.ng = ngm[pick];
    CSh.g_bytes = (uint32_t)(P.jstride[CSh.gax] * (int64_t)h->esz);
    CSh.w_bytes = (uint32_t)(P.jstride[5 - CSh.gax] * (int64_t)h->esz);
    void *d = nullptr;
    {
        std::vector<int32_t> wgw;
        st = colcoop_plan<T>(h, plan[pick], cells[pick], wgw);      // fills the plans' staged-row offsets
        if (st) return st;
        if (h->cs_coop_epl) {
            st = upload(h, wgw, &d);
            if (st) return st;
            CSh.wg = (const int32_t *)d;
        }
        CSh.coop = h->cs_coop ? h->cs_coop_epl : 0;
    }
    st = upload(h, plan[pick], &d);
    if (st) return st;
    CSh.plan = (const int32_t *)d;
    CSh.npre_col = npre_col;
    CSh.step_uniform = step_uniform ? 1 : 0;
    CSh.ncu = ncu;
    CSh.s1_bytes = (uint32_t)(P.jstride[1] * (int64_t)h->esz);
    st = colsweep_map(h, plan[pick]);
    if (st) return st;
    {
        bool dok = false;
        st = colsweep_dpp_ok<T>(h, &dok);
        if (st) return st;
        CSh.dpp = (dok && h->cs_dpp) ? 1 : 0;
    }
    colsweep_split(h);
    st = dev_alloc(h, sizeof(DColSweep), &d);
    if (st) return st;
    h->dcs = (DColSweep *)d;
    st = colsweep_upload(h);
    if (st) return st;
    h->cs_state = 1;
    return HJB_OK;
}

int ensure_colsweep(Handle *h) {
    if (h->dtype == HJB_F64) { if (h->cs_state < 0) h->cs_state = 0; return HJB_OK; }   // float32 arithmetic only
    return ensure_colsweep_t<float>(h);
}

// K9 applies when, for every state and control, each axis' interpolation cell is the state's own cell or the one
// below (clamped to the grid): then J_k at a state depends on J_{k+1} within +-1 cell only.  Checked on the host
// from the variant-5 tables (small: 2-D problems only).
template <typename T>
int examine_tile2d_t(Handle *h) {
    h->tile2d = 0;
    const DParams &P = h->hp;
    if (P.D != 2 || h->j_elems != h->n_owned || !h->tabled_ok || h->hp.model) return HJB_OK;
    // few controls only (the launch-bound channels this is for), and tables small enough that checking them on the
    // host costs nothing next to the sweep
    if (h->nU > 64 || h->dom_entries[0] + h->dom_entries[1] > ((int64_t)1 << 24)) return HJB_OK;
    int st = ensure_tabled(h);
    if (st) return st;
    for (int a = 0; a < 2; ++a) {
        std::vector<TabEntry<T>> tab((size_t)h->dom_entries[a]);
        HIP_TRY(h, hipMemcpy(tab.data(), h->htb.ax[a].tab, tab.size() * sizeof(TabEntry<T>), hipMemcpyDeviceToHost));
        // entry index -> this axis' state index: strides of the table domain
        const DTabled::Axis &A = h->htb.ax[a];
        const int na = P.n[a];
        if (A.sstride[a] == 0) return HJB_OK;               // x_next_a does not depend on x_a: not a local problem
        // walk every entry: its axis-a index is (e / sstride[a]) % n[a] because domains are dense column-major
        for (int64_t e = 0; e < h->dom_entries[a]; ++e) {
            const int i = (int)((e / A.sstride[a]) % na);
            const int lo = std::max(i - 1, 0), hi = std::min(i, na - 2);
            if (tab[(size_t)e].cell < lo || tab[(size_t)e].cell > hi) return HJB_OK;
        }
    }
    h->tile2d = 1;
    if (P.C == 1 && h->nU <= kTileMaxU) {          // the cached form: its per-(state, control) plan, built once
        const int64_t ne = h->n_owned * h->nU;
        void *d = nullptr;
        st = dev_alloc(h, (size_t)ne * sizeof(TilePlan<T>), &d);
        if (st) return st;
        (void)stage_tile2d_plan(h->dtype, h->dp, h->dtb, d, ne);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, sync_setup());
        h->tile_plan = d;
    }
    return HJB_OK;
}

int examine_tile2d(Handle *h) {
    return h->dtype != HJB_F64 ? examine_tile2d_t<float>(h) : examine_tile2d_t<double>(h);
}

int launch_tile2d(Handle *h, const void *dJn, void *dJo, void *didx, int K, hipStream_t st) {
    const DParams &P = h->hp;
    StageArgs a;
    a.grid = (unsigned)(((P.n[0] + kTileX - 1) / kTileX) * ((P.n[1] + kTileY - 1) / kTileY));
    a.block = 256;
    a.st = st;
    a.dtype = h->dtype;
    a.D = P.D;
    a.dp = h->dp;
    a.dtb = h->dtb;
    a.Jn = dJn;
    a.Jo = dJo;
    a.idx = didx;
    (void)stage_tile2d(a, h->tile_plan, K);      // the cached form when its plan exists
    HIP_TRY(h, hipGetLastError());
    return HJB_OK;
}

void choose_launch(Handle *h) {
    if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    // few states x many controls (Kirk): one wave per state, controls across lanes
    const bool want_split = h->nU >= 64 && h->n_owned < 512 * 1024 && !h->tab64;
    // variant 7 (column sweep) wants what variant 6 wants - long axis-0 rows on a large grid - plus its own structure;
    // its plan is built here (never inside a launch: launches may be under graph capture)
    bool cs_auto = false;
    if (h->hp.D == 4 && h->hp.C == 1 && !h->hp.model && (h->forced_variant == 7 || (h->forced_variant < 0 && h->row_auto && !h->packed_mode && !h->nested_ok && !want_split))) {
        if (h->cs_state < 0 && ensure_colsweep(h) != HJB_OK) h->cs_state = 0;
        cs_auto = h->cs_state == 1;
    }
    h->variant = h->forced_variant >= 0 ? h->forced_variant
                                        : (h->packed_mode ? 4 : (h->nested_ok ? 1 : (want_split ? 3 : (cs_auto ? 7 : (h->row_auto ? 6 : (h->tabled_ok ? 5 : 0))))));
    if (h->hp.model) h->variant = 4;
    if (h->dtype == HJB_F16S && h->variant >= 1 && h->variant <= 3)     // float16 J storage: variants 0, 4, 5, 6, 7 only
        h->variant = h->forced_variant >= 0 ? h->forced_variant : (cs_auto ? 7 : (h->row_auto ? 6 : (h->tabled_ok ? 5 : 0)));
    if (h->variant == 7 && h->cs_state != 1) h->variant = h->row_ok ? 6 : (h->tabled_ok ? 5 : 0);
    if (h->tab64 && h->variant < 5) h->variant = 5;       // float64-built tables: the table-driven kernels only (tabled_ok holds)
    // float64 cost terms: the tabled kernel, or the column sweep in its usual cost shape (state terms + one control term)
    if (h->cost64 && !(h->variant == 5 || (h->variant == 7 && h->hcs.ncu == 1 && h->hp.n_cost_prefix > 0 && !h->hcs.coop))) h->variant = 5;
    // build the variant 5/6 tables now (never inside a launch: launches may be under graph capture)
    h->launch_status = HJB_OK;
    if ((h->variant == 5 || h->variant == 6) && (h->launch_status = ensure_tabled(h)) != HJB_OK) {
        // a float64-table handle never falls back to a kernel that evaluates the float32 copies of its terms:
        // it keeps its variant and every launch reports the build's status (hjb_create fails on it)
        if (!h->tab64 && !h->cost64) h->variant = 0;          // ... and neither does a float64-cost handle (kernels 5 / 7 only)
    }
    h->block = 256;
    h->split_j_in_lds = (size_t)h->j_elems * h->esz <= 64 * 1024;
    // the control-split kernel with J staged in LDS: eight waves share one copy of J (Kirk: 40 KB), so four workgroups fill a CU's 32
    // wave slots instead of half of them (Kirk's default problem 16.1 -> 13.4 ms per 199 stages: profiles/r06_xcd_shares_and_spans.log)
    if (h->variant == 3 && h->split_j_in_lds) h->block = 512;
    const int per_block = h->variant == 2 ? 512 : (h->variant == 3 ? h->block / 64 : 256);   // states per workgroup pass (variant 4: 256)
    int64_t blocks = (h->n_owned + per_block - 1) / per_block;
    // A launch smaller than the work walks it in grid-sized spans.  Equally long spans: a short last span runs on part of the chip
    // (Solver_attitude.run's 5199 chunks as 4096 + 1103: 3.63 ms per 19 stages; as 2 x 2600: 2.53), and the kernels that give XCD x
    // the x-th contiguous share of every span (kernels_packed2.h, kernels_tabled.h) would hand a short one to the first XCDs alone.
    auto spans_of = [](int64_t work, int64_t cap) {
        if (work <= cap) return work;
        const int64_t spans = (work + cap - 1) / cap;
        return std::min<int64_t>(cap, ((work + spans - 1) / spans + 7) / 8 * 8);      // (a multiple of 8: the window modes' walk asks for it)
    };
    // (the control-split kernel keeps its 1024 workgroups: one wave per state and few states - Kirk's 2500 blocks as 3 x 840 ran 19.3 ms
    // per 199 stages against 16.3 with a short last span that overlaps the tail of the one before)
    // The table kernel takes its whole grid as ONE span where its 32-bit form allows (XCD x then sweeps one contiguous eighth of the
    // grid: 13M states 0.671 -> 0.630 ms, Solver_attitude.run in the reference's order 13.7 -> 12.9 ms per 19 stages; 2e8 states: equal)
    const int64_t cap = h->variant == 5 ? kTab32MaxThreads / 256 : 256 * 16;
    h->grid = h->variant == 3 ? (int)std::min<int64_t>(blocks, h->block == 512 ? 2048 : 1024) : (int)spans_of(blocks, cap);
    if (h->variant == 6) {       // one wave per (64-state chunk of a) grid row, four waves per workgroup
        const int64_t n0 = h->hp.n[0];
        const int64_t items = (h->n_owned / n0) * ((n0 + 63) / 64);
        h->grid = (int)spans_of((items + 3) / 4, 1 << 20);        // (one span where it can: C4 in the reference's order 6.49 -> 6.15 ms per stage)
    }
    if (h->variant == 7) {       // one wave per (chunk of axis 0, i2, i3) column; workgroup b serves XCD b % 8
        const DParams &P = h->hp;
        const int lanes = h->hcs.dpp ? kCsDppLanes : 64;
        const int64_t chunks = (P.n[0] + lanes - 1) / lanes;
        const int64_t nwax = P.n[5 - h->hcs.gax];
        int64_t most = 0;
        const int64_t nfull = h->hcs.xcd_win ? P.n[h->hcs.gax] : nwax;      // the axis every XCD walks in full
        for (int x = 0; x < 8; ++x) most = std::max<int64_t>(most, (int64_t)h->hcs.xcd_cnt[x] * chunks * nfull * h->hcs.split);
        h->grid = (int)(8 * ((most + 3) / 4));
        h->cc_grid = 0;
        if (h->hcs.coop && !h->hcs.xcd_win) {       // cooperative form: one workgroup of kCcW waves per (group-axis index, 64-state chunk, kCcW columns)
            const int64_t c64 = (P.n[0] + 63) / 64, nblk = (nwax + kCcW - 1) / kCcW;
            int64_t mostc = 0;
            for (int x = 0; x < 8; ++x) mostc = std::max<int64_t>(mostc, (int64_t)h->hcs.xcd_cnt[x] * c64 * nblk);
            h->cc_grid = (int)(8 * mostc);
        }
    }
    if (h->variant == 4 && uniwin_active(h)) h->grid = h->uw_grid;      // K15: one generation of workgroups
    if (h->grid < 1) h->grid = 1;
}

// One stage: the handle's variant on (dJn -> dJo, didx).  The kernels live in translation units of their own
// (stage_*.hip behind hjbdp_launch.h); this is the only place that knows which family serves which variant.
int launch_stage(Handle *h, const void *dJn, void *dJo, void *didx, hipStream_t st) {
    const int D = h->hp.D;
    const bool f32 = h->dtype != HJB_F64;              // float32 arithmetic (J stored as float32 or binary16)
    const bool same = h->dtype != HJB_F16S;            // J stored in the arithmetic type
    StageArgs a;
    a.grid = (unsigned)h->grid;
    a.block = (unsigned)h->block;
    a.st = st;
    a.dtype = h->dtype;
    a.D = D;
    a.dp = h->dp;
    a.dn = h->dn;
    a.dtb = h->dtb;
    a.dcs = h->dcs;
    a.Jn = dJn;
    a.Jo = dJo;
    a.idx = didx;
    int miss = 0;
    if (h->tab64 && (h->variant < 5 || h->launch_status != HJB_OK))
        return fail(h, h->launch_status != HJB_OK ? h->launch_status : HJB_E_UNSUPPORTED,
                    "table_dtype HJB_TAB_F64 is served by the table-driven kernels only (variant %d, table build status %d)", h->variant, h->launch_status);
    if (h->cost64 && h->variant != 5 && h->variant != 7)
        return fail(h, HJB_E_UNSUPPORTED, "cost_dtype HJB_COST_F64 is served by stage kernels 5 and 7 only (variant %d)", h->variant);
    switch (h->variant) {
        case 7: {
            if (!h->dtb || !h->dcs) return fail(h, HJB_E_DEVICE, "variant 7 plan missing");
            if (!f32) return fail(h, HJB_E_UNSUPPORTED, "variant 7 is float32 arithmetic only");
            const bool fastcost = h->hcs.ncu == 1 && h->hp.n_cost_prefix > 0;    // state terms + one control term
            if (h->cost64 && !fastcost) return fail(h, HJB_E_UNSUPPORTED, "variant 7 sums float64 cost terms in its usual cost shape only");
            const int costform = h->cost64 ? 2 : (fastcost ? 1 : 0);
            // cooperative form: its staging loads are 16 bytes wide (a J pointer handed in unaligned runs the other form)
            if (!h->cost64 && h->hcs.coop && h->cc_grid > 0 && h->hcs.ng <= kCcNCG && ((uintptr_t)dJn & 15u) == 0) {
                a.grid = (unsigned)h->cc_grid;
                miss = stage_colcoop(a, h->hcs.gax, h->hcs.ng, fastcost);
            } else {
                miss = stage_colsweep(a, h->hcs.gax, h->hcs.ng, costform, h->hcs.dpp != 0);
            }
            if (miss) return fail(h, HJB_E_DEVICE, "variant 7: %d groups", h->hcs.ng);
            break;
        }
        case 6: {
            if (!h->dtb) return fail(h, HJB_E_DEVICE, "variant 6 tables missing");
            const bool lean = h->row_lean && h->row_lean_ok && !h->htb.ax[0].has_ctrl;
            const size_t tsz = f32 ? 4 : 8;
            const size_t lean_wave = (((size_t)h->nU * 4 + 15) & ~(size_t)15) + (((size_t)h->nU * (D - 1 + kLeanMaxCu) * tsz + 15) & ~(size_t)15);
            a.lds = lean ? 4 * lean_wave + (size_t)h->nU * 12 : 0;
            miss = stage_rowwise(a, lean);
            break;
        }
        case 5:
            if (!h->dtb) return fail(h, HJB_E_DEVICE, "variant 5 tables missing");
            a.idx32 = h->tabled_i32 && h->tabled_i32_on && (int64_t)a.grid * a.block <= kTab32MaxThreads;
            miss = stage_tabled(a);
            break;
        case 4:
            if (!f32) return fail(h, HJB_E_UNSUPPORTED, "variant 4 is float32 only");
            if (uniwin_active(h)) {                  // modes 7 / 8 (K15, kernels_uniwin.h)
                a.duw = h->duw;
                a.block = (unsigned)h->huw.block;
                if (h->huw.counters) HIP_TRY(h, hipMemsetAsync(h->huw.counters, 0, 8 * 16 * sizeof(uint32_t), st));   // (a memset node under capture)
                a.lds = h->uw_lds + h->lds_pad;
                miss = stage_uniwin(a, h->hp.model != 0);
                break;
            }
            a.lds = h->packed2_lds + h->lds_pad;
            miss = stage_packed2(a, h->packed_pre);
            break;
        case 3:
            if (!same) return fail(h, HJB_E_UNSUPPORTED, "variant 3 does not support float16 J storage (use 0, 4 or 5)");
            a.lds = h->split_j_in_lds ? (size_t)h->j_elems * h->esz : 0;
            miss = stage_ctrlsplit(a, h->split_j_in_lds);
            break;
        case 2:
            if (!same) return fail(h, HJB_E_UNSUPPORTED, "variant 2 does not support float16 J storage (use 0, 4 or 5)");
            if (!f32) return fail(h, HJB_E_UNSUPPORTED, "variant 2 is float32 only");
            a.lds = h->packed_lds;
            miss = stage_packed(a);
            break;
        case 1:
            if (!same) return fail(h, HJB_E_UNSUPPORTED, "variant 1 does not support float16 J storage (use 0, 4 or 5)");
            a.lds = h->nested_lds;
            miss = stage_nested(a, h->nested_fast);
            break;
        case 0:
            miss = stage_generic(a);
            break;
        default:    // never fall through to the generic kernel silently
            return fail(h, HJB_E_DEVICE, "internal: kernel variant %d was not dispatched", h->variant);
    }
    if (miss) return fail(h, HJB_E_UNSUPPORTED, "variant %d has no kernel for D=%d, dtype %d", h->variant, D, h->dtype);
    HIP_TRY(h, hipGetLastError());
    return HJB_OK;
}

int ensure_work(Handle *h) {
    if (h->dJ[0]) return HJB_OK;
    for (int i = 0; i < 2; ++i) {
        int st = dev_alloc(h, (size_t)h->j_elems * h->esz, &h->dJ[i]);
        if (st) return st;
        HIP_TRY(h, hipMemset(h->dJ[i], 0, (size_t)h->j_elems * h->esz));
    }
    void *d = nullptr;
    int st = dev_alloc(h, (size_t)h->n_owned * h->idx_bytes, &d);
    if (st) return st;
    h->d_idx = (char *)d;
    st = dev_alloc(h, sizeof(double) * 2 * kReduceBlocks, &d);
    if (st) return st;
    h->d_partials = (double *)d;
    st = dev_alloc(h, sizeof(double) * 2, &d);
    if (st) return st;
    h->d_sums = (double *)d;
    return HJB_OK;
}

int check_status(Handle *h, hipStream_t st) {
    int32_t flag = 0;
    HIP_TRY(h, hipMemcpyAsync(&flag, h->d_status, sizeof flag, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    if (flag) {
        HIP_TRY(h, hipMemsetAsync(h->d_status, 0, sizeof(int32_t), st));
        return fail(h, HJB_E_HALO, "a next-state query left the slab's halo (halo_lo=%d halo_hi=%d; tables imply lo=%d hi=%d)",
                    h->hp.halo_lo, h->nplanes - h->hp.n[h->hp.D - 1] - h->hp.halo_lo, h->halo_need_lo, h->halo_need_hi);
    }
    return HJB_OK;
}

// ---- probe block (Dynamic_Solver.m:212-219) -------------------------------------------------------------------
int make_probe(Handle *h, const hjb_probe *pb, DProbe *out) {
    if (h->hp.model) return fail(h, HJB_E_UNSUPPORTED, "the probe block is not available for problems with a state model");
    if (h->tab64) return fail(h, HJB_E_UNSUPPORTED, "the probe block reports float32 next states; not available with table_dtype HJB_TAB_F64");
    if (h->cost64) return fail(h, HJB_E_UNSUPPORTED, "the probe block reports the float32 stage cost; not available with cost_dtype HJB_COST_F64");
    memset(out, 0, sizeof *out);
    int64_t B = 1;
    for (int a = 0; a < h->hp.D; ++a) {
        if (pb->lo[a] < 0 || pb->hi[a] > h->prob.n[a] || pb->lo[a] >= pb->hi[a])
            return fail(h, HJB_E_INVALID, "probe block [%d, %d) on axis %d of %d points (the reference's taps 50:55, 52:57 need dx >= 57, "
                        "Dynamic_Solver.m:213)", pb->lo[a], pb->hi[a], a, h->prob.n[a]);
        out->lo[a] = pb->lo[a];
        out->ext[a] = pb->hi[a] - pb->lo[a];
        B *= out->ext[a];
    }
    for (int c = 0; c < HJB_MAX_C; ++c) {
        const int mc = c < h->hp.C ? h->prob.m[c] : 1;
        if (c < h->hp.C && (pb->control[c] < 0 || pb->control[c] >= mc))
            return fail(h, HJB_E_INVALID, "probe control index %d on control dim %d of %d levels (the reference's tap 105 needs du >= 105)",
                        pb->control[c], c, mc);
        out->control[c] = c < h->hp.C ? pb->control[c] : 0;
    }
    if (B > ((int64_t)1 << 24)) return fail(h, HJB_E_INVALID, "probe block of %lld states is too large", (long long)B);
    out->B = B;
    return HJB_OK;
}

int launch_probe(Handle *h, const DProbe &pr, const void *dJn, hipStream_t st) {
    dim3 g((unsigned)std::min<int64_t>((pr.B + 255) / 256, 4096)), b(256);
#define HJB_LAUNCH_PROBE(TT, TTJ)                                                                                     \
    switch (h->hp.D) {                                                                                                \
        case 1: hipLaunchKernelGGL((k_probe<TT, TTJ, 1>), g, b, 0, st, h->dp, pr, (const TTJ *)dJn); break;            \
        case 2: hipLaunchKernelGGL((k_probe<TT, TTJ, 2>), g, b, 0, st, h->dp, pr, (const TTJ *)dJn); break;            \
        case 3: hipLaunchKernelGGL((k_probe<TT, TTJ, 3>), g, b, 0, st, h->dp, pr, (const TTJ *)dJn); break;            \
        case 4: hipLaunchKernelGGL((k_probe<TT, TTJ, 4>), g, b, 0, st, h->dp, pr, (const TTJ *)dJn); break;            \
        case 5: hipLaunchKernelGGL((k_probe<TT, TTJ, 5>), g, b, 0, st, h->dp, pr, (const TTJ *)dJn); break;            \
        default: hipLaunchKernelGGL((k_probe<TT, TTJ, 6>), g, b, 0, st, h->dp, pr, (const TTJ *)dJn); break;           \
    }
    if (h->dtype == HJB_F16S) { HJB_LAUNCH_PROBE(float, _Float16) }
    else if (h->dtype == HJB_F32) { HJB_LAUNCH_PROBE(float, float) }
    else { HJB_LAUNCH_PROBE(double, double) }
#undef HJB_LAUNCH_PROBE
    HIP_TRY(h, hipGetLastError());
    return HJB_OK;
}

// ---- what the other units call of the templates above

// ---- K15 (kernels_uniwin.h): variant 4's three-plane window modes on chunks that share their rate axes -------------------
// Applies when, beyond modes 5 / 6, nothing the level axes and the last axis need depends on the state-only axes: their tables'
// domains, the last axis' state terms and its inner term (Solver_attitude.m:423-425: the next rates are functions of the rates
// and the torque).  The plan is built here; `uniwin_auto` says whether the usual shape holds on (nearly) every point.
void uniwin_tiles(Handle *h) {
    DUniwin &U = h->huw;
    U.block = h->uw_block == 64 ? 64 : 256;
    U.cpp = (int32_t)((U.inner + U.block - 1) / U.block);
    {
        size_t ot_floats = 0;
        for (int i = HJB_MAX_D; i < HJB_MAX_D + 2; ++i)
            if (h->hn.ot[i].present) ot_floats = std::max<size_t>(ot_floats, (size_t)h->hn.ot[i].lds_off + (size_t)h->hn.ot[i].lds_len);
        h->uw_lds = (size_t)27 * U.block * 4 + ot_floats * 4 + 16;
    }
    auto lg = [](int n, int most) { int l = 0; while (l < most && (1 << l) < n) ++l; return l; };
    int lA = 3, lB = 2, lC = 2;
    if (h->uw_tile > 0) { lA = h->uw_tile & 7; lB = (h->uw_tile >> 3) & 7; lC = (h->uw_tile >> 6) & 7; }
    U.lA = lg(U.nA, lA);
    U.lB = lg(U.nB, lB);
    U.lC = lg(U.nC, lC);
    U.ntA = (U.nA + (1 << U.lA) - 1) >> U.lA;
    U.ntB = (U.nB + (1 << U.lB) - 1) >> U.lB;
    U.ntC = (U.nC + (1 << U.lC) - 1) >> U.lC;
    U.tile_chunks = (uint32_t)U.cpp << (U.lA + U.lB + U.lC);
    const uint64_t nv = (uint64_t)U.tile_chunks * (uint64_t)U.ntA * (uint64_t)U.ntB * (uint64_t)U.ntC;
    U.n_v = (uint32_t)std::min<uint64_t>(nv, 0xfffffff0u);
    // the launch: as many workgroups as the device holds at once (a persistent walk: a second generation would run alone)
    int occ = stage_uniwin_occupancy(h->dtype, h->hp.D, h->hp.model != 0, U.block, h->uw_lds + h->lds_pad);
    if (occ < 1) occ = U.block == 64 ? 16 : 4;
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    int64_t g = (int64_t)occ * cus;
    g = std::min<int64_t>(g, (int64_t)((U.n_v + 7) / 8) * 8);
    h->uw_grid = (int)std::max<int64_t>(8, g - (g & 7));
}

static int setup_uniwin(Handle *h, const hjb_problem *p) {
    const DParams &P = h->hp;
    const DNested &N = h->hn;
    const int D = p->D, C = p->C;
    h->uniwin_ok = h->uniwin_auto = false;
    if (!(h->packed_mode && (h->packed_pre == 5 || h->packed_pre == 6)) || !h->dn) return HJB_OK;
    if (C != 3 || D < 4 || D > 6) return HJB_OK;
    if (!(N.m_in == kUwIn || N.m_in == kUwIn - 1) || N.m_o0 > kUwMaxO || N.m_o1 > kUwMaxO) return HJB_OK;
    const int NP = D - 3, AX_A = D - 3, AX_B = D - 2;
    const uint32_t so_bits = (1u << NP) - 1u;             // the state-only dims
    for (int d = 0; d < NP; ++d)
        if (N.at[AX_A].sstride[d] != 0 || N.at[AX_B].sstride[d] != 0) return HJB_OK;
    if (N.at[AX_B].c0 != 0 || !N.at[AX_A].tab || !N.at[AX_B].tab) return HJB_OK;
    for (int k = 0; k <= N.ax_kin && k < p->n_next_terms[D - 1]; ++k)
        if (p->next_terms[D - 1][k].mask & so_bits) return HJB_OK;
    if (N.n_ax_in != 1 || N.n_cost_in != 1 || N.in[kMaxInAx].lds_slot < 0) return HJB_OK;
    int64_t inner = 1;
    for (int a = 0; a < NP; ++a) inner *= p->n[a];
    const int64_t n_points = h->n_owned / inner;
    if (inner < 128 || inner >= ((int64_t)1 << 30) || n_points >= ((int64_t)1 << 24) || h->inner >= ((int64_t)1 << 31)) return HJB_OK;
    size_t ot_floats = 0;
    for (int i = HJB_MAX_D; i < HJB_MAX_D + 2; ++i)
        if (N.ot[i].present) {
            if (N.ot[i].lds_off < 0) return HJB_OK;
            ot_floats = std::max<size_t>(ot_floats, (size_t)N.ot[i].lds_off + (size_t)N.ot[i].lds_len);
        }
    if ((size_t)27 * 256 * 4 + ot_floats * 4 + 16 > 64 * 1024) return HJB_OK;
    DUniwin &U = h->huw;
    memset(&U, 0, sizeof U);
    U.n_points = (int32_t)n_points;
    U.inner = (int32_t)inner;
    U.nA = p->n[AX_A];
    U.nB = p->n[AX_B];
    U.nC = P.n[D - 1];                                     // owned planes
    U.cl1_per_o0 = (N.ot[HJB_MAX_D + 1].present && N.ot[HJB_MAX_D + 1].c0 != 0) ? 1 : 0;
    if ((int64_t)U.nA * U.nB * U.nC != n_points) return HJB_OK;
    void *plan = nullptr, *cnt = nullptr;
    int st = dev_alloc(h, (size_t)n_points * kUwRec * sizeof(int32_t), &plan);
    if (!st) st = dev_alloc(h, sizeof(int32_t), &cnt);
    if (st) return st;
    HIP_TRY(h, hipMemset(cnt, 0, sizeof(int32_t)));
    if (stage_uniwin_plan(D, h->dp, h->dn, (int32_t *)plan, (int)n_points, U.nA, U.nB, (int32_t *)cnt)) return HJB_OK;
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, sync_setup());
    int32_t n_slow = 0;
    HIP_TRY(h, hipMemcpy(&n_slow, cnt, sizeof n_slow, hipMemcpyDeviceToHost));
    h->uniwin_slow = n_slow;
    U.plan = (const int32_t *)plan;
    {   // the per-XCD claim counters of the chunk walk (kernels_uniwin.h): 8 x one 64-byte line, zeroed before every launch
        void *ctr = nullptr;
        st = dev_alloc(h, 8 * 16 * sizeof(uint32_t), &ctr);
        if (st) return st;
        HIP_TRY(h, hipMemset(ctr, 0, 8 * 16 * sizeof(uint32_t)));
        U.counters = (uint32_t *)ctr;
    }
    uniwin_tiles(h);
    void *du = nullptr;
    st = dev_alloc(h, sizeof(DUniwin), &du);
    if (st) return st;
    h->duw = (DUniwin *)du;
    HIP_TRY(h, hipMemcpy(h->duw, &U, sizeof U, hipMemcpyHostToDevice));
    h->uniwin_ok = true;
    h->uniwin_auto = (int64_t)n_slow * 50 <= n_points;     // at most 2 % of the points on the slow path
    return HJB_OK;
}

int build_handle(Handle *h, const hjb_problem *p) {
    const int st = p->dtype != HJB_F64 ? build<float>(h, p) : build<double>(h, p);
    if (st) return st;
    return setup_uniwin(h, p);
}


void halo_of_problem(const hjb_problem *p, bool tab64, int *lo, int *hi) {
    if (p->dtype != HJB_F64) halo_from_terms<float>(p, tab64, lo, hi);
    else halo_from_terms<double>(p, tab64, lo, hi);
}

int colsweep_dpp_ok_f32(Handle *h, bool *ok) { return colsweep_dpp_ok<float>(h, ok); }

}  // namespace hjbhost
