// hjbdp.hip - C ABI of libhjbdp (include/hjbdp.h): problem upload, stage-kernel
// dispatch, the backward sweep loop and the early-stop monitor.
// gfx950 (MI355X) only; no CPU fallback - without a HIP device every compute
// entry point returns HJB_E_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <chrono>
#include <climits>
#include <cstring>
#include <string>
#include <type_traits>
#include <mutex>
#include <shared_mutex>
#include <vector>

#include "../../include/hjbdp.h"
#include "hjbdp_dev.h"
#include "hjbdp_launch.h"
#include "kernels_generic.h"
#include "kernels_nested.h"
#include "kernels_packed.h"
#include "kernels_packed2.h"
#include "kernels_ctrlsplit.h"
#include "kernels_lookup.h"
#include "kernels_tabled.h"
#include "kernels_rowwise.h"
#include "kernels_tile2d.h"
#include "kernels_colsweep.h"
#include "kernels_colcoop.h"
#include <atomic>
#include "kernels_probe.h"
#include "kernels_prep_mfma.h"
#include "kernels_reduce.h"

struct ncclUniqueIdBytes { char internal[128]; };      // rccl.h ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
#include "kernels_devmem.h"

using namespace hjb;

namespace {

constexpr int kGraphStages = 32;   // even: a replay starts and ends in dJ[0]

thread_local std::string g_last_error;
std::atomic<int> g_test_fail_tab64_scratch{0};      // hjb_test_hook("fail_tab64_scratch", v): fault injection for the tests
std::atomic<int> g_test_fail_tabled_alloc{0};       // hjb_test_hook("fail_tabled_alloc", v): the (cell, t) table allocation fails
std::atomic<int> g_test_rccl_only_env{0};           // hjb_test_hook("rccl_only_env", v): the loader tries $HJBDP_RCCL_LIB only (a host without librccl)
// Handles may be driven from different host threads (one thread per handle).  HIP stream capture is fragile
// against "unsafe" calls made elsewhere in the process while it records (device-wide synchronisation, synchronous
// copies, allocation): a capture takes this lock exclusively, every such call takes it shared.  Kernel launches,
// graph launches and waits on a handle's own stream need no lock and overlap freely.
static std::shared_mutex g_capture_mu;

struct Handle {
    hjb_problem prob{};  // scalar fields only (pointers are not kept)
    int device = 0;
    int dtype = HJB_F32;
    size_t esz = 4;
    int64_t n_owned = 0, nU = 0, j_elems = 0, inner = 0;
    int nplanes = 0, plane0 = 0;
    DParams hp{};                 // host copy of the device params
    DParams *dp = nullptr;        // device params
    std::vector<void *> allocs;   // every device allocation (freed in destroy)
    int32_t *d_status = nullptr;
    // work buffers (lazy)
    void *dJ[2] = {nullptr, nullptr};
    char *d_idx = nullptr;        // argmin labels of the owned states, idx_bytes each
    int idx_bytes = 4;            // hjb_problem.idx_dtype resolved: 4 (int32), 1 (uint8) or 2 (uint16)
    bool tab64 = false;           // hjb_problem.table_dtype == HJB_TAB_F64: (cell, t) tables built in float64 from float64 terms
    bool cost64 = false;          // hjb_problem.cost_dtype == HJB_COST_F64: cost terms float64, summed in double, one rounding per backup
    DParams *dp64 = nullptr;      // ... the float64 shadow of the axes (knots, 1/dx, next-state terms) the table build reads
    double *d_partials = nullptr;  // monitor reduction scratch
    double *d_sums = nullptr;      // [2]: sum J, sum idx
    DNested hn{};                 // variant 1 (control-nested) parameters
    DNested *dn = nullptr;
    bool nested_ok = false;
    bool nested_fast = false;
    int packed_mode = 0;          // variant 2 eligibility
    bool split_j_in_lds = false;  // variant 3: whole J buffer staged in LDS
    // launch-bound sweeps: the ping-pong stage loop captured once into a hipGraph of kGraphStages launches
    hipStream_t stream = nullptr;
    hipGraphExec_t gexec = nullptr;
    int gexec_variant = -1;
    bool gexec_tiled = false;
    bool use_graph = true;
    bool monitor_single = false;  // option "monitor_single" (see hjb_solve_opts.monitor_single)
    size_t packed_lds = 0;
    size_t packed2_lds = 0;       // variant 4 (two controls per packed op)
    void *tile_plan = nullptr;    // K9 cached form: per (state, control) stage-invariant record (k_tile2d_plan)
    int tile2d = -1;              // K9 (several stages per launch, kernels_tile2d.h): -1 not examined yet, 0 no, 1 yes
    int use_temporal = 1;         // option "temporal": 0 off, 1 when applicable, 2 required (hjb_solve fails otherwise)
    bool row_ok = false;          // variant 6 (one wave per grid row) applies
    bool row_auto = false;        // ... and is chosen automatically
    bool row_lean_ok = false;     // variant 6: the lean form applies (kernels_rowwise.h)
    bool row_lean = true;         // option "row_lean"
    int packed_pre = 0;           // variant 4 contraction mode (kernels_packed2.h MODE): 0 plain, 1 C2 shape, 2 state-only axes first
    bool window3_ok = false;      // modes 2 / 3 qualify for the three-plane window (modes 5 / 6); option "window_planes" switches
    size_t lds_pad = 0;           // extra dynamic LDS per workgroup (occupancy tuning)
    bool tabled_ok = false;       // variant 5: per-axis (cell, t) tables for every axis (built on first use)
    uint32_t dom_mask[HJB_MAX_D] = {0};
    int64_t dom_entries[HJB_MAX_D] = {0};
    DTabled htb{};
    DTabled *dtb = nullptr;
    size_t nested_lds = 0;
    // every stage-invariant (cell, weight) table of this handle: rebuilt by option "prep_mfma" (timing / equality tests)
    struct PrepRec { int axis; int kind; const int32_t *dsz_d; std::vector<int32_t> dsz; int64_t n; void *tab; };
    std::vector<PrepRec> preps;
    bool inline_axis0 = true;     // allow mode 1's axis 0 without a table (see build)
    bool axis0_inline = false;    // ... in effect: N.at[0].tab is null
    uint32_t axis0_dom = 0;       // its broadcast domain and entry count, should the table be wanted after all
    int64_t axis0_nent = 0;
    int prep_mfma = 0;            // 1: tables were built with v_mfma_f32_32x32x2_f32 where the axis' terms allow it
    int prep_mfma_axes = 0;       // ... number of tables the MFMA form applied to in the last rebuild
    double prep_us = 0;           // device time of the last rebuild of all tables
    int cs_state = -1;            // variant 7 (column sweep, kernels_colsweep.h): -1 not examined, 0 does not apply, 1 plan built
    DColSweep hcs{};
    DColSweep *dcs = nullptr;
    int cs_xcd_mod = 0;           // option "cs_xcd_mod": 0 = automatic (see colsweep_map)
    int cs_dpp = 1;               // option "cs_dpp": allow the DPP form of variant 7 when the axis-0 cells permit it
    int cs_rows_mid = 0;          // corner rows per step the mid-grid column needs (get_option "cs_rows")
    int cs_xcd_axis = 0;          // option "cs_xcd_axis": 0 = the XCDs split the group axis, 1 = the window axis
    int cs_split = 0;             // option "cs_split": parts a column is swept in (0 = automatic, see colsweep_split)
    int cs_coop = 0;              // option "cs_coop": allow the cooperative form (kernels_colcoop.h) where it applies
    std::vector<double> cs_cu64;  // cost_dtype F64: the control term of the cost in float64, per control (plan building)
    int cs_coop_why = 0;          // why it does not: 1 groups, 2 axis 1 sees the window axis, 3 n0 / storage, 4 cells, 5 window knots, 6 axis-0 knots
    int cs_coop_epl = 0;          // ... it applies: elements per staging load (0 = does not apply)
    int cc_grid = 0;              // its launch grid
    int variant = 0;
    int launch_status = HJB_OK;   // status of the table build inside choose_launch
    int forced_variant = -1;
    int block = 256, grid = 0;
    int halo_need_lo = 0, halo_need_hi = 0;
    std::string err;
};

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

#define HIP_TRY(h, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, HJB_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

template <typename T>
int upload(Handle *h, const std::vector<T> &v, void **out) {
    void *d = nullptr;
    HIP_TRY(h, hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    h->allocs.push_back(d);
    if (!v.empty()) HIP_TRY(h, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = d;
    return HJB_OK;
}

int dev_alloc(Handle *h, size_t bytes, void **out) {
    void *d = nullptr;
    HIP_TRY(h, hipMalloc(&d, std::max<size_t>(bytes, 16)));
    h->allocs.push_back(d);
    *out = d;
    return HJB_OK;
}

int64_t term_elems(const hjb_problem *p, uint32_t mask) {
    int64_t s = 1;
    for (int d = 0; d < p->D + p->C; ++d)
        if (mask & (1u << d)) s *= (d < p->D) ? p->n[d] : p->m[d - p->D];
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
    HIP_TRY(h, hipDeviceSynchronize());
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
            HIP_TRY(h, hipDeviceSynchronize());
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
            const hipError_t e1 = hipDeviceSynchronize();
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
    HIP_TRY(h, hipDeviceSynchronize());
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

int rebuild_tables(Handle *h, bool mfma) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(h, hipEventCreate(&e0));
    HIP_TRY(h, hipEventCreate(&e1));
    HIP_TRY(h, hipDeviceSynchronize());
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
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
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
    HIP_TRY(h, hipMemcpy(h->dcs, &CSh, sizeof(DColSweep), hipMemcpyHostToDevice));
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
        HIP_TRY(h, hipDeviceSynchronize());
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
    const int per_block = h->variant == 2 ? 512 : (h->variant == 3 ? 4 : 256);   // states per workgroup pass (variant 4: 256)
    int64_t blocks = (h->n_owned + per_block - 1) / per_block;
    h->grid = (int)std::min<int64_t>(blocks, h->variant == 3 ? 1024 : 256 * 16);
    if (h->variant == 6) {       // one wave per (64-state chunk of a) grid row, four waves per workgroup
        const int64_t n0 = h->hp.n[0];
        const int64_t items = (h->n_owned / n0) * ((n0 + 63) / 64);
        h->grid = (int)std::min<int64_t>((items + 3) / 4, 256 * 16);
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
            miss = stage_tabled(a);
            break;
        case 4:
            if (!f32) return fail(h, HJB_E_UNSUPPORTED, "variant 4 is float32 only");
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

}  // namespace

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
static int analyse_problem(const hjb_problem *p, int *idx_bytes_out, int64_t *n_states_out, int *halo_lo, int *halo_hi) {
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
    for (int a = 0; a < p->D; ++a) {
        if (p->n[a] < 2) return fail(nullptr, HJB_E_INVALID, "n[%d]=%d < 2", a, p->n[a]);
        if (!p->knots[a]) return fail(nullptr, HJB_E_INVALID, "knots[%d] is null", a);
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
        }
        nS *= p->n[a];
    }
    for (int c = 0; c < p->C; ++c) {
        if (p->m[c] < 1) return fail(nullptr, HJB_E_INVALID, "m[%d]=%d < 1", c, p->m[c]);
        nU *= p->m[c];
    }
    if (nU >= (int64_t)1 << 31) return fail(nullptr, HJB_E_UNSUPPORTED, "too many controls");
    if (p->n_cost_terms < 1 || p->n_cost_terms > HJB_MAX_TERMS) return fail(nullptr, HJB_E_INVALID, "n_cost_terms=%d", p->n_cost_terms);
    for (int k = 0; k < p->n_cost_terms; ++k)
        if (!p->cost_terms[k].data || (p->cost_terms[k].mask >> G)) return fail(nullptr, HJB_E_INVALID, "cost term %d: bad mask/data", k);
    for (int k = 0; k < p->n_cost_terms; ++k)
        if (term_elems(p, p->cost_terms[k].mask) >= ((int64_t)1 << 31)) return fail(nullptr, HJB_E_UNSUPPORTED, "cost term %d has >= 2^31 elements", k);
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
        else if (p->dtype != HJB_F64) halo_from_terms<float>(p, tab64, halo_lo, halo_hi);
        else halo_from_terms<double>(p, tab64, halo_lo, halo_hi);
    }
    return HJB_OK;
}

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
    int st = p->dtype != HJB_F64 ? build<float>(h, p) : build<double>(h, p);
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
    info->lds_bytes = h->variant == 4 ? (int32_t)h->packed2_lds : h->variant == 2 ? (int32_t)h->packed_lds
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
            const int cst = colsweep_dpp_ok<float>(h, &dok);
            if (cst) return cst;
            h->hcs.dpp = (dok && h->cs_dpp) ? 1 : 0;
            colsweep_split(h);
            HIP_TRY(h, hipMemcpy(h->dcs, &h->hcs, sizeof(DColSweep), hipMemcpyHostToDevice));
            choose_launch(h);
        }
        return HJB_OK;
    }
    if (!strcmp(key, "cs_split")) {                                  // variant 7: parts a column is swept in (0 = automatic)
        if (value < 0 || value > 64) return fail(h, HJB_E_INVALID, "%s out of range", key);
        h->cs_split = (int)value;
        if (h->cs_state == 1) {
            colsweep_split(h);
            HIP_TRY(h, hipMemcpy(h->dcs, &h->hcs, sizeof(DColSweep), hipMemcpyHostToDevice));
            choose_launch(h);
        }
        return HJB_OK;
    }
    if (!strcmp(key, "cs_coop")) {                                   // 0: variant 7 runs one wave per column (testing)
        h->cs_coop = value != 0;
        if (h->cs_state == 1) {
            h->hcs.coop = h->cs_coop ? h->cs_coop_epl : 0;
            HIP_TRY(h, hipMemcpy(h->dcs, &h->hcs, sizeof(DColSweep), hipMemcpyHostToDevice));
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
            HIP_TRY(h, hipMemcpy(h->dcs, &h->hcs, sizeof(DColSweep), hipMemcpyHostToDevice));
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
            HIP_TRY(h, hipMemcpy(h->dcs, &h->hcs, sizeof(DColSweep), hipMemcpyHostToDevice));
            choose_launch(h);
        }
        return HJB_OK;
    }
    if (!strcmp(key, "lds_pad")) {
        if (value < 0 || value > 128 * 1024) return fail(h, HJB_E_INVALID, "lds_pad out of range");
        h->lds_pad = (size_t)value;
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
        if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }   // the captured launches are the other form
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
    else if (!strcmp(key, "packed2_mode")) *value = h->packed_mode ? h->packed_pre : -1;   // variant 4's contraction mode (kernels_packed2.h), -1: not eligible
    else if (!strcmp(key, "idx_bytes")) *value = h->idx_bytes;
    else if (!strcmp(key, "temporal")) *value = h->use_temporal;
    else if (!strcmp(key, "chunk_order")) *value = h->dn ? h->hn.chunk_order : 0;
    else if (!strcmp(key, "row_lean")) *value = h->row_lean ? 1 : 0;
    else if (!strcmp(key, "lds_pad")) *value = (int64_t)h->lds_pad;
    else if (!strcmp(key, "cs_xcd_mod")) *value = h->cs_xcd_mod;
    else if (!strcmp(key, "cs_xcd_axis")) *value = h->cs_xcd_axis;
    else if (!strcmp(key, "cs_split")) *value = h->variant == 7 ? h->hcs.split : 0;       // the value in effect
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
    SOLVE_TRY(hipDeviceSynchronize());   // the sweep runs on the handle's own stream from here
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

// ---- flat builder API (MATLAB loadlibrary/calllib cannot marshal hjb_problem) ---------------------------------------
struct hjb_builder_s {
    hjb_problem p{};
    std::vector<std::vector<double>> knots;
    std::vector<std::vector<unsigned char>> blobs;   // owned copies of every term / model table
    std::string err;
};

static int bfail(hjb_builder b, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (b) b->err = buf;
    g_last_error = buf;
    return code;
}

const char *hjb_problem_last_error(hjb_builder b) { return b ? b->err.c_str() : g_last_error.c_str(); }

int32_t hjb_problem_new(int32_t D, int32_t C, const int32_t *n, const int32_t *m, int32_t dtype, int32_t index_base,
                        hjb_builder *out) {
    if (!out || !n || !m) return bfail(nullptr, HJB_E_INVALID, "null argument");
    *out = nullptr;
    if (D < 1 || D > HJB_MAX_D) return bfail(nullptr, HJB_E_UNSUPPORTED, "D=%d not in 1..%d", D, HJB_MAX_D);
    if (C < 1 || C > HJB_MAX_C) return bfail(nullptr, HJB_E_UNSUPPORTED, "C=%d not in 1..%d", C, HJB_MAX_C);
    if (dtype != HJB_F32 && dtype != HJB_F64 && dtype != HJB_F16S) return bfail(nullptr, HJB_E_UNSUPPORTED, "dtype %d", dtype);
    if (index_base != 0 && index_base != 1) return bfail(nullptr, HJB_E_INVALID, "index_base must be 0 or 1");
    for (int a = 0; a < D; ++a) if (n[a] < 2) return bfail(nullptr, HJB_E_INVALID, "n[%d]=%d < 2", a, n[a]);
    for (int c = 0; c < C; ++c) if (m[c] < 1) return bfail(nullptr, HJB_E_INVALID, "m[%d]=%d < 1", c, m[c]);
    hjb_builder b = new hjb_builder_s();
    b->p.D = D; b->p.C = C; b->p.dtype = dtype; b->p.index_base = index_base;
    for (int a = 0; a < D; ++a) b->p.n[a] = n[a];
    for (int c = 0; c < C; ++c) b->p.m[c] = m[c];
    b->knots.resize((size_t)D);
    *out = b;
    return HJB_OK;
}

int32_t hjb_problem_set_knots(hjb_builder b, int32_t axis, const double *knots, int32_t len) {
    if (!b || !knots) return bfail(b, HJB_E_INVALID, "null argument");
    if (axis < 0 || axis >= b->p.D) return bfail(b, HJB_E_INVALID, "axis %d not in 0..%d", axis, b->p.D - 1);
    if (len != b->p.n[axis]) return bfail(b, HJB_E_INVALID, "axis %d has %d grid points, %d knots given", axis, b->p.n[axis], len);
    b->knots[(size_t)axis].assign(knots, knots + len);
    return HJB_OK;
}

// bytes per element of a term array the caller hands in: next-state terms are float64 under table_dtype HJB_TAB_F64
static size_t term_esz(const hjb_problem &p, bool next_term) {
    return (p.dtype == HJB_F64 || (next_term && p.table_dtype == HJB_TAB_F64) || (!next_term && p.cost_dtype == HJB_COST_F64)) ? 8 : 4;
}

int32_t hjb_problem_set_cost_type(hjb_builder b, int32_t cost_dtype) {
    if (!b) return bfail(b, HJB_E_INVALID, "null builder");
    if (cost_dtype != HJB_COST_DEFAULT && cost_dtype != HJB_COST_F64) return bfail(b, HJB_E_INVALID, "cost_dtype %d", cost_dtype);
    if (cost_dtype == HJB_COST_F64 && b->p.dtype == HJB_F64) return bfail(b, HJB_E_INVALID, "cost_dtype HJB_COST_F64 is for float32 arithmetic");
    if (cost_dtype != b->p.cost_dtype && b->p.n_cost_terms)
        return bfail(b, HJB_E_INVALID, "set the cost dtype before adding cost terms (it is their element type)");
    b->p.cost_dtype = cost_dtype;
    return HJB_OK;
}

int32_t hjb_problem_set_types(hjb_builder b, int32_t idx_dtype, int32_t table_dtype) {
    if (!b) return bfail(b, HJB_E_INVALID, "null builder");
    if (idx_dtype < HJB_IDX_I32 || idx_dtype > HJB_IDX_AUTO) return bfail(b, HJB_E_INVALID, "idx_dtype %d", idx_dtype);
    if (table_dtype != HJB_TAB_DEFAULT && table_dtype != HJB_TAB_F64) return bfail(b, HJB_E_INVALID, "table_dtype %d", table_dtype);
    if (table_dtype == HJB_TAB_F64 && b->p.dtype == HJB_F64) return bfail(b, HJB_E_INVALID, "table_dtype HJB_TAB_F64 is for float32 arithmetic");
    if (table_dtype != b->p.table_dtype)
        for (int a = 0; a < b->p.D; ++a)
            if (b->p.n_next_terms[a]) return bfail(b, HJB_E_INVALID, "set the table dtype before adding next-state terms (it is their element type)");
    b->p.idx_dtype = idx_dtype;
    b->p.table_dtype = table_dtype;
    return HJB_OK;
}

static int add_term(hjb_builder b, hjb_term *slot, uint32_t mask, const void *data, int64_t count, const char *what, bool next_term) {
    if (!data) return bfail(b, HJB_E_INVALID, "%s: null data", what);
    if (mask >> (b->p.D + b->p.C)) return bfail(b, HJB_E_INVALID, "%s: mask 0x%x names a grid dim >= %d", what, mask, b->p.D + b->p.C);
    const int64_t need = term_elems(&b->p, mask);
    if (count != need) return bfail(b, HJB_E_INVALID, "%s: mask 0x%x spans %lld elements, %lld given", what, mask, (long long)need, (long long)count);
    const size_t esz = term_esz(b->p, next_term);
    b->blobs.emplace_back((const unsigned char *)data, (const unsigned char *)data + (size_t)count * esz);
    slot->mask = mask;
    slot->reserved = 0;
    slot->data = nullptr;          // bound in hjb_create_from (the vectors may still move)
    return HJB_OK;
}

int32_t hjb_problem_add_next_term(hjb_builder b, int32_t axis, uint32_t mask, const void *data, int64_t count) {
    if (!b) return bfail(b, HJB_E_INVALID, "null builder");
    if (axis < 0 || axis >= b->p.D) return bfail(b, HJB_E_INVALID, "axis %d not in 0..%d", axis, b->p.D - 1);
    if (b->p.n_next_terms[axis] >= HJB_MAX_TERMS) return bfail(b, HJB_E_UNSUPPORTED, "more than %d terms for axis %d", HJB_MAX_TERMS, axis);
    hjb_term *slot = &b->p.next_terms[axis][b->p.n_next_terms[axis]];
    const int st = add_term(b, slot, mask, data, count, "next term", true);
    if (st) return st;
    slot->reserved = (uint32_t)b->blobs.size();          // 1-based blob number until hjb_create_from binds the pointer
    ++b->p.n_next_terms[axis];
    return HJB_OK;
}

int32_t hjb_problem_add_cost_term(hjb_builder b, uint32_t mask, const void *data, int64_t count) {
    if (!b) return bfail(b, HJB_E_INVALID, "null builder");
    if (b->p.n_cost_terms >= HJB_MAX_TERMS) return bfail(b, HJB_E_UNSUPPORTED, "more than %d cost terms", HJB_MAX_TERMS);
    hjb_term *slot = &b->p.cost_terms[b->p.n_cost_terms];
    const int st = add_term(b, slot, mask, data, count, "cost term", false);
    if (st) return st;
    slot->reserved = (uint32_t)b->blobs.size();
    ++b->p.n_cost_terms;
    return HJB_OK;
}

int32_t hjb_problem_set_slab(hjb_builder b, int32_t slab_begin, int32_t slab_end, int32_t halo_lo, int32_t halo_hi) {
    if (!b) return bfail(b, HJB_E_INVALID, "null builder");
    b->p.slab_begin = slab_begin; b->p.slab_end = slab_end; b->p.halo_lo = halo_lo; b->p.halo_hi = halo_hi;
    return HJB_OK;
}

int32_t hjb_problem_set_model(hjb_builder b, int32_t model, double model_h, const void *t0, const void *t1,
                              const void *t2, const void *t3) {
    if (!b) return bfail(b, HJB_E_INVALID, "null builder");
    if (model == HJB_MODEL_NONE) { b->p.model = HJB_MODEL_NONE; return HJB_OK; }
    if (model != HJB_MODEL_QUAT_EULER321) return bfail(b, HJB_E_INVALID, "model %d", model);
    if (b->p.D != 6 || b->p.C != 3 || b->p.dtype == HJB_F64) return bfail(b, HJB_E_UNSUPPORTED, "HJB_MODEL_QUAT_EULER321 needs D=6, C=3, float32 arithmetic");
    const void *t[4] = {t0, t1, t2, t3};
    const size_t ne = (size_t)b->p.n[0] * b->p.n[1] * b->p.n[2];
    for (int i = 0; i < 4; ++i) {
        if (!t[i]) return bfail(b, HJB_E_INVALID, "model table %d is null", i);
        b->blobs.emplace_back((const unsigned char *)t[i], (const unsigned char *)t[i] + ne * 4);
        b->p.model_tables[i] = (const void *)(uintptr_t)b->blobs.size();     // blob number, bound in hjb_create_from
    }
    b->p.model = model;
    b->p.model_h = model_h;
    return HJB_OK;
}

// Relabel the state axes of a problem under construction: new axis i = old axis order[i].  Pure bookkeeping - which
// axis is "last" decides which stage kernel applies, the order of the 1-D lerps and the axis a multi-GPU run shards -
// but the term arrays are stored over their dims in ascending order, so a term over several state dims is transposed.
// The caller permutes its own arrays the same way: MATLAB `permute(J, order + 1)` in, `ipermute` out.
int32_t hjb_problem_permute_axes(hjb_builder b, const int32_t *order) {
    if (!b || !order) return bfail(b, HJB_E_INVALID, "null argument");
    hjb_problem &p = b->p;
    const int D = p.D, C = p.C;
    if (p.model != HJB_MODEL_NONE) return bfail(b, HJB_E_UNSUPPORTED, "a problem with a state model has a fixed axis labelling");
    if (p.slab_begin || p.slab_end || p.halo_lo || p.halo_hi) return bfail(b, HJB_E_INVALID, "permute the axes before setting a slab");
    int new_of_old[HJB_MAX_G], seen = 0;
    for (int i = 0; i < D; ++i) {
        if (order[i] < 0 || order[i] >= D || (seen >> order[i]) & 1) return bfail(b, HJB_E_INVALID, "order is not a permutation of 0..%d", D - 1);
        seen |= 1 << order[i];
        new_of_old[order[i]] = i;
    }
    for (int c = 0; c < C; ++c) new_of_old[D + c] = D + c;
    int gn[HJB_MAX_G];                                           // old grid sizes of all dims
    for (int a = 0; a < D; ++a) gn[a] = p.n[a];
    for (int c = 0; c < C; ++c) gn[D + c] = p.m[c];
    auto remap = [&](hjb_term &t, bool next_term) {
        const size_t esz = term_esz(p, next_term);
        int od[HJB_MAX_G], k = 0;                                // the term's dims, ascending (old labels) = its storage order
        for (int d = 0; d < D + C; ++d) if ((t.mask >> d) & 1u) od[k++] = d;
        uint32_t nm = 0;
        for (int i = 0; i < k; ++i) nm |= 1u << new_of_old[od[i]];
        // storage position of old dim od[i] in the new array = rank of its new label
        int pos[HJB_MAX_G];
        for (int i = 0; i < k; ++i) {
            pos[i] = 0;
            for (int j = 0; j < k; ++j) pos[i] += new_of_old[od[j]] < new_of_old[od[i]];
        }
        bool same = true;
        for (int i = 0; i < k; ++i) same = same && pos[i] == i;
        t.mask = nm;
        if (same) return;
        std::vector<unsigned char> &blob = b->blobs[t.reserved - 1];
        std::vector<unsigned char> out(blob.size());
        int64_t nstride[HJB_MAX_G], sz[HJB_MAX_G];               // stride (elements) of old dim i in the new array
        for (int i = 0; i < k; ++i) sz[i] = gn[od[i]];
        for (int i = 0; i < k; ++i) {
            nstride[i] = 1;
            for (int j = 0; j < k; ++j) if (pos[j] < pos[i]) nstride[i] *= sz[j];
        }
        int64_t idx[HJB_MAX_G] = {0}, total = 1;
        for (int i = 0; i < k; ++i) total *= sz[i];
        for (int64_t e = 0; e < total; ++e) {                    // e walks the old array in storage order
            int64_t o = 0;
            for (int i = 0; i < k; ++i) o += idx[i] * nstride[i];
            memcpy(&out[(size_t)o * esz], &blob[(size_t)e * esz], esz);
            for (int i = 0; i < k; ++i) { if (++idx[i] < sz[i]) break; idx[i] = 0; }
        }
        blob.swap(out);
    };
    for (int a = 0; a < D; ++a)
        for (int k = 0; k < p.n_next_terms[a]; ++k) remap(p.next_terms[a][k], true);
    for (int k = 0; k < p.n_cost_terms; ++k) remap(p.cost_terms[k], false);
    hjb_problem q = p;
    std::vector<std::vector<double>> kn((size_t)D);
    for (int i = 0; i < D; ++i) {
        const int o = order[i];
        q.n[i] = p.n[o];
        q.n_next_terms[i] = p.n_next_terms[o];
        for (int k = 0; k < HJB_MAX_TERMS; ++k) q.next_terms[i][k] = p.next_terms[o][k];
        kn[(size_t)i] = b->knots[(size_t)o];
    }
    p = q;
    b->knots.swap(kn);
    return HJB_OK;
}

// A labelling of the state axes under which a faster stage kernel applies.  D = 4 with one control dim: the one under
// which the column-sweep stage kernel applies (the pos-att shape: D = 4, one control
// dim, two axes whose next value involves neither the control nor each other's state dim, two that involve their own
// pair of dims and the control only), found from the terms' masks alone; of the two control-driven axes the one the
// controls move less - the larger (next - own) range of its control-only terms over its mean knot spacing goes first -
// comes last (its halo is the narrower one for a multi-GPU run).  order_out[i] = the present axis that becomes axis i;
// *found = 0 and the identity when no labelling qualifies (or the present one already does).
int32_t hjb_problem_suggest_order(hjb_builder b, int32_t *order_out, int32_t *found) {
    if (!b || !order_out || !found) return bfail(b, HJB_E_INVALID, "null argument");
    const hjb_problem &p = b->p;
    const int D = p.D;
    for (int i = 0; i < D; ++i) order_out[i] = i;
    *found = 0;
    if (p.model != HJB_MODEL_NONE) return HJB_OK;
    uint32_t dom[HJB_MAX_D];
    for (int a = 0; a < D; ++a) {
        dom[a] = 0;
        for (int k = 0; k < p.n_next_terms[a]; ++k) dom[a] |= p.next_terms[a][k].mask;
    }
    if (D != 4 || p.C != 1) {
        // The general rule of the fast kernels (control-nested, packed): axes the controls do not drive first (they are
        // contracted once per state), then the driven axes in the order of the control loops, the axis of the innermost
        // control dim last (Solver_attitude.m's (w1, w2, w3, yaw, pitch, roll) becomes (yaw, pitch, roll, w1, w2, w3):
        // 4.4 instead of 28 ms on the reference grid).  A stable sort: axes of equal rank keep their order.
        int key[HJB_MAX_D];
        for (int a = 0; a < D; ++a) {
            key[a] = 0;
            for (int c = 0; c < p.C; ++c) if ((dom[a] >> (D + c)) & 1u) key[a] = 1 + c;
        }
        int ord[HJB_MAX_D];
        for (int i = 0; i < D; ++i) ord[i] = i;
        std::stable_sort(ord, ord + D, [&](int x, int y) { return key[x] < key[y]; });
        bool ident = true;
        for (int i = 0; i < D; ++i) ident = ident && ord[i] == i;
        if (!ident) {
            for (int i = 0; i < D; ++i) order_out[i] = ord[i];
            *found = 1;
        }
        return HJB_OK;
    }
    const uint32_t cbit = 1u << 4;
    auto spread = [&](int a) -> double {                         // range of the axis' control-only terms, in mean knot spacings
        double lo = 0, hi = 0;
        for (int k = 0; k < p.n_next_terms[a]; ++k) {
            const hjb_term &t = p.next_terms[a][k];
            if (t.mask != cbit) continue;
            const std::vector<unsigned char> &bl = b->blobs[t.reserved - 1];
            double tl = 0, th = 0;
            for (int u = 0; u < p.m[0]; ++u) {
                const double v = term_esz(p, true) == 8 ? ((const double *)bl.data())[u] : (double)((const float *)bl.data())[u];
                tl = u == 0 ? v : std::min(tl, v);
                th = u == 0 ? v : std::max(th, v);
            }
            lo += tl; hi += th;
        }
        const std::vector<double> &kn = b->knots[(size_t)a];
        const double h = kn.size() > 1 ? (kn.back() - kn.front()) / (double)(kn.size() - 1) : 1.0;
        return h > 0 ? (hi - lo) / h : 0.0;
    };
    int best[4] = {0, 1, 2, 3};
    double best_score = 0;
    int perm[4] = {0, 1, 2, 3};
    do {
        // new axis i = old axis perm[i]; an old dim d carries new label pos(d)
        int pos[4];
        for (int i = 0; i < 4; ++i) pos[perm[i]] = i;
        auto relabel = [&](uint32_t m) { uint32_t r = m & cbit; for (int d = 0; d < 4; ++d) if ((m >> d) & 1u) r |= 1u << pos[d]; return r; };
        const uint32_t d0 = relabel(dom[perm[0]]), d1 = relabel(dom[perm[1]]), d2 = relabel(dom[perm[2]]), d3 = relabel(dom[perm[3]]);
        if ((d0 & (cbit | 2u)) || (d1 & (cbit | 1u)) || (d2 & 3u) || (d3 & 3u)) continue;
        for (size_t a = 0; a < 4; ++a) if (b->knots[a].empty()) return bfail(b, HJB_E_INVALID, "set the knots before asking for an axis order");
        // prefer: the less-moved control axis last; then the labelling closest to the present one
        double score = spread(perm[2]) - spread(perm[3]);
        int moved = 0;
        for (int i = 0; i < 4; ++i) moved += perm[i] != i;
        score -= 1e-6 * moved;
        if (!*found || score > best_score) { best_score = score; for (int i = 0; i < 4; ++i) best[i] = perm[i]; *found = 1; }
    } while (std::next_permutation(perm, perm + 4));
    if (*found) {
        bool ident = true;
        for (int i = 0; i < 4; ++i) { order_out[i] = best[i]; ident = ident && best[i] == i; }
        if (ident) *found = 0;
    }
    return HJB_OK;
}

static int builder_bind(hjb_builder b, hjb_problem *out) {      // the builder's problem with its pointers bound
    hjb_problem p = b->p;
    for (int a = 0; a < p.D; ++a) {
        if (b->knots[(size_t)a].empty()) return bfail(b, HJB_E_INVALID, "knots of axis %d were not set", a);
        p.knots[a] = b->knots[(size_t)a].data();
        for (int k = 0; k < p.n_next_terms[a]; ++k) {
            p.next_terms[a][k].data = b->blobs[p.next_terms[a][k].reserved - 1].data();
            p.next_terms[a][k].reserved = 0;
        }
    }
    for (int k = 0; k < p.n_cost_terms; ++k) {
        p.cost_terms[k].data = b->blobs[p.cost_terms[k].reserved - 1].data();
        p.cost_terms[k].reserved = 0;
    }
    if (p.model != HJB_MODEL_NONE)
        for (int i = 0; i < 4; ++i) p.model_tables[i] = b->blobs[(size_t)(uintptr_t)b->p.model_tables[i] - 1].data();
    *out = p;
    return HJB_OK;
}

int32_t hjb_create_from(hjb_builder b, int32_t device, hjb_handle *out) {
    if (!b || !out) return bfail(b, HJB_E_INVALID, "null argument");
    hjb_problem p;
    const int st0 = builder_bind(b, &p);
    if (st0) return st0;
    const int st = hjb_create(&p, device, out);
    if (st) b->err = g_last_error;
    return st;
}

int32_t hjb_problem_free(hjb_builder b) {
    delete b;
    return HJB_OK;
}

int32_t hjb_solve_flat(hjb_handle h, int32_t n_stages, int32_t monitor_period, double monitor_tol, const void *terminal,
                       void *J_final, void *idx_final, void *J_stages, void *idx_stages, int32_t *stages_done,
                       int32_t *stopped_early, double *sweep_ms) {
    hjb_solve_opts o{};
    o.n_stages = n_stages;
    o.monitor_period = monitor_period;
    o.monitor_tol = monitor_tol;
    o.terminal = terminal;
    o.J_final = J_final;
    o.idx_final = idx_final;
    o.J_stages = J_stages;
    o.idx_stages = idx_stages;
    hjb_result r{};
    const int st = hjb_solve(h, &o, &r);
    if (stages_done) *stages_done = r.stages_done;
    if (stopped_early) *stopped_early = r.stopped_early;
    if (sweep_ms) *sweep_ms = r.sweep_ms;
    return st;
}

int32_t hjb_get_info_flat(hjb_handle h, int64_t *out8) {
    if (!out8) return fail((Handle *)h, HJB_E_INVALID, "null argument");
    hjb_info i{};
    const int st = hjb_get_info(h, &i);
    if (st) return st;
    out8[0] = i.n_states; out8[1] = i.n_controls; out8[2] = i.j_elems; out8[3] = i.kernel_variant;
    out8[4] = i.lds_bytes; out8[5] = i.grid; out8[6] = i.halo_needed_lo; out8[7] = i.halo_needed_hi;
    return HJB_OK;
}

// ---- single-process multi-GPU sweep ------------------------------------------------------------------------------
// The reference's stage loop (pos-att/Solver_pos_att.m:270-286) over a grid partitioned along its LAST state axis into
// one slab per device.  Per stage and slab: the halo planes of J_{k+1} are copied from the neighbouring slabs
// (hipMemcpyPeerAsync on a copy stream; xGMI between GPUs), the INTERIOR planes - whose next states stay inside the
// owned planes - are backed up while the copies are in flight, the two boundary strips afterwards.  Interior and strips
// are slab handles over the same buffers (a slab handle sees planes [begin - halo_lo, end + halo_hi)).
struct hjb_multi_s {
    struct Slab {
        int device = 0, begin = 0, end = 0, hlo = 0, hhi = 0;
        Handle *whole = nullptr;             // owns the J buffers (dJ[0], dJ[1]) and idx
        Handle *part[3] = {nullptr, nullptr, nullptr};     // interior, low strip, high strip (null: no split)
        int64_t part_row0[3] = {0, 0, 0};    // first plane of the part's view inside the slab's J buffer
        int64_t part_own0[3] = {0, 0, 0};    // first owned plane of the part, relative to `begin`
        hipStream_t sc = nullptr, sx = nullptr;
        hipStream_t ss[2] = {nullptr, nullptr};              // the two boundary strips run beside the interior
        hipEvent_t done[2] = {nullptr, nullptr}, halo[2] = {nullptr, nullptr};
        hipEvent_t fork = nullptr, sdone[2] = {nullptr, nullptr};
    };
    std::vector<Slab> slabs;
    int need_lo = 0, need_hi = 0, nl = 0, dtype = HJB_F32;
    int64_t inner = 0;
    size_t esz = 4, isz = 4;                 // bytes per J element / per argmin label
    std::string err;
};

static int mfail(hjb_multi m, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (m) m->err = buf;
    g_last_error = buf;
    return code;
}

const char *hjb_multi_last_error(hjb_multi m) { return m ? m->err.c_str() : g_last_error.c_str(); }

int32_t hjb_destroy_multi(hjb_multi m) {
    if (!m) return HJB_OK;
    for (auto &S : m->slabs) {
        (void)hipSetDevice(S.device);
        (void)hipDeviceSynchronize();
        for (int i = 0; i < 2; ++i) {
            if (S.done[i]) (void)hipEventDestroy(S.done[i]);
            if (S.halo[i]) (void)hipEventDestroy(S.halo[i]);
            if (S.sdone[i]) (void)hipEventDestroy(S.sdone[i]);
            if (S.ss[i]) (void)hipStreamDestroy(S.ss[i]);
        }
        if (S.fork) (void)hipEventDestroy(S.fork);
        if (S.sc) (void)hipStreamDestroy(S.sc);
        if (S.sx) (void)hipStreamDestroy(S.sx);
        for (int i = 0; i < 3; ++i) if (S.part[i]) (void)hjb_destroy((hjb_handle)S.part[i]);
        if (S.whole) (void)hjb_destroy((hjb_handle)S.whole);
    }
    delete m;
    return HJB_OK;
}

int32_t hjb_create_multi(const hjb_problem *p, int32_t n_dev, const int32_t *devices, hjb_multi *out) {
    if (!p || !devices || !out) return mfail(nullptr, HJB_E_INVALID, "null argument");
    *out = nullptr;
    if (n_dev < 1 || n_dev > 64) return mfail(nullptr, HJB_E_INVALID, "n_dev=%d", n_dev);
    if (p->slab_begin || p->slab_end || p->halo_lo || p->halo_hi) return mfail(nullptr, HJB_E_INVALID, "hjb_create_multi partitions the grid itself: pass the whole problem");
    if (p->D < 1 || p->D > HJB_MAX_D) return mfail(nullptr, HJB_E_UNSUPPORTED, "D=%d", p->D);
    const int nl = p->n[p->D - 1];
    if (n_dev > nl) return mfail(nullptr, HJB_E_INVALID, "more devices (%d) than planes of the last axis (%d)", n_dev, nl);
    // the halo the tables imply and the label width: host arithmetic on the last axis' terms - no whole-grid handle, no
    // whole-grid tables (a problem whose slabs fit must not be refused because the whole grid would not)
    hjb_info pin{};
    int st;
    {
        int ib = 4, hl = 0, hh = 0;
        int64_t ns = 0;
        st = analyse_problem(p, &ib, &ns, &hl, &hh);
        if (st) return st;
        pin.idx_bytes = ib; pin.n_states = ns; pin.halo_needed_lo = hl; pin.halo_needed_hi = hh;
    }
    hjb_multi m = new hjb_multi_s();
    m->need_lo = pin.halo_needed_lo;
    m->need_hi = pin.halo_needed_hi;
    m->nl = nl;
    m->dtype = p->dtype;
    m->esz = p->dtype == HJB_F16S ? 2 : (p->dtype == HJB_F32 ? 4 : 8);
    m->inner = pin.n_states / nl;
    m->isz = (size_t)pin.idx_bytes;
    m->slabs.resize((size_t)n_dev);
    const int base = nl / n_dev, rem = nl % n_dev;
    int b = 0;
    for (int i = 0; i < n_dev; ++i) {
        auto &S = m->slabs[(size_t)i];
        S.device = devices[i];
        S.begin = b;
        S.end = b + base + (i < rem ? 1 : 0);
        b = S.end;
        S.hlo = std::min(m->need_lo, S.begin);
        S.hhi = std::min(m->need_hi, nl - S.end);
    }
    for (int i = 0; i < n_dev; ++i) {       // a halo must come from the immediate neighbour only
        const auto &S = m->slabs[(size_t)i];
        if ((i > 0 && S.hlo > m->slabs[(size_t)i - 1].end - m->slabs[(size_t)i - 1].begin) ||
            (i + 1 < n_dev && S.hhi > m->slabs[(size_t)i + 1].end - m->slabs[(size_t)i + 1].begin)) {
            (void)hjb_destroy_multi(m);
            return mfail(nullptr, HJB_E_INVALID, "halo (%d/%d planes) wider than a neighbouring slab: use fewer devices or relabel the "
                         "state axes so that the last axis moves less", m->need_lo, m->need_hi);
        }
    }
    auto make = [&](int dev, int sb, int se, int hl, int hh, Handle **hout) {
        hjb_problem q = *p;
        if (n_dev > 1) { q.slab_begin = sb; q.slab_end = se; q.halo_lo = hl; q.halo_hi = hh; }
        hjb_handle h = nullptr;
        const int s2 = hjb_create(&q, dev, &h);
        *hout = (Handle *)h;
        return s2;
    };
    for (int i = 0; i < n_dev && !st; ++i) {
        auto &S = m->slabs[(size_t)i];
        st = make(S.device, S.begin, S.end, S.hlo, S.hhi, &S.whole);
        if (st) break;
        const int lo_w = S.hlo ? m->need_lo : 0, hi_w = S.hhi ? m->need_hi : 0, owned = S.end - S.begin;
        if (n_dev > 1 && owned - lo_w - hi_w >= 1 && (lo_w || hi_w)) {
            const int view0 = S.begin - S.hlo;
            auto sub = [&](int k, int sb, int se, int hl, int hh) {
                S.part_row0[k] = (sb - hl) - view0;
                S.part_own0[k] = sb - S.begin;
                return make(S.device, sb, se, hl, hh, &S.part[k]);
            };
            st = sub(0, S.begin + lo_w, S.end - hi_w, std::min(m->need_lo, lo_w), std::min(m->need_hi, hi_w));
            if (!st && lo_w) st = sub(1, S.begin, S.begin + lo_w, S.hlo, std::min(m->need_hi, S.end - (S.begin + lo_w)));
            if (!st && hi_w) st = sub(2, S.end - hi_w, S.end, std::min(m->need_lo, (S.end - hi_w) - S.begin), S.hhi);
        }
        if (st) break;
        if (hipSetDevice(S.device) != hipSuccess) { st = mfail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d)", S.device); break; }
        {
            std::shared_lock<std::shared_mutex> lk(g_capture_mu);
            st = ensure_work(S.whole);
        }
        if (st) break;
        bool ok = hipStreamCreateWithFlags(&S.sc, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&S.sx, hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; k < 2 && ok; ++k)
            ok = hipEventCreateWithFlags(&S.done[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&S.halo[k], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&S.sdone[k], hipEventDisableTiming) == hipSuccess && hipStreamCreateWithFlags(&S.ss[k], hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&S.fork, hipEventDisableTiming) == hipSuccess;
        if (!ok) { st = mfail(nullptr, HJB_E_DEVICE, "stream / event creation failed on device %d", S.device); break; }
        for (int j = 0; j < n_dev; ++j)          // direct peer copies where the platform allows them (errors: staged copies still work)
            if (devices[j] != S.device) { int can = 0; if (hipDeviceCanAccessPeer(&can, S.device, devices[j]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(devices[j], 0); }
        (void)hipGetLastError();
    }
    if (st) {
        const std::string keep = g_last_error;
        (void)hjb_destroy_multi(m);
        g_last_error = keep;
        return st;
    }
    *out = m;
    return HJB_OK;
}

int32_t hjb_multi_slab_info(hjb_multi m, int32_t slab, int32_t *begin, int32_t *end, int32_t *halo_lo, int32_t *halo_hi,
                            int32_t *split, int32_t *kernel_variant) {
    if (!m || slab < 0 || slab >= (int)m->slabs.size()) return mfail(m, HJB_E_INVALID, "slab %d", slab);
    const auto &S = m->slabs[(size_t)slab];
    if (begin) *begin = S.begin;
    if (end) *end = S.end;
    if (halo_lo) *halo_lo = S.hlo;
    if (halo_hi) *halo_hi = S.hhi;
    if (split) *split = S.part[0] ? 1 : 0;
    if (kernel_variant) *kernel_variant = (S.part[0] ? S.part[0] : S.whole)->variant;
    return HJB_OK;
}

int32_t hjb_multi_set_option(hjb_multi m, const char *key, int64_t value) {
    if (!m || !key) return mfail(m, HJB_E_INVALID, "null argument");
    for (auto &S : m->slabs) {
        Handle *hs[4] = {S.whole, S.part[0], S.part[1], S.part[2]};
        for (Handle *h : hs)
            if (h) {
                const int st = hjb_set_option((hjb_handle)h, key, value);
                if (st) return mfail(m, st, "%s", hjb_last_error((hjb_handle)h));
            }
    }
    return HJB_OK;
}

int32_t hjb_solve_multi(hjb_multi m, const hjb_solve_opts *o, hjb_result *res) {
    if (!m || !o) return mfail(m, HJB_E_INVALID, "null argument");
    if (o->n_stages < 1) return mfail(m, HJB_E_INVALID, "n_stages=%d", o->n_stages);
    if (o->probe)
        return mfail(m, HJB_E_UNSUPPORTED, "hjb_solve_multi takes no probe block (use one device, or drive the slabs yourself)");
    if (o->monitor_single && o->monitor_period > 0)
        return mfail(m, HJB_E_UNSUPPORTED, "monitor_single (a float32 running sum in one fixed order over the whole grid) is for one device; "
                     "hjb_solve_multi adds exact float64 sums over the slabs");
    const bool every_stage = o->progress && o->progress_every_stage;
    const int n = (int)m->slabs.size();
    const int64_t inner = m->inner;
    const size_t esz = m->esz, plane_b = (size_t)inner * esz;
#define MULTI_TRY(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return mfail(m, HJB_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    // terminal cost into buffer 0 of every slab (halo planes are filled by the first exchange)
    for (auto &S : m->slabs) {
        MULTI_TRY(hipSetDevice(S.device));
        char *J0 = (char *)S.whole->dJ[0];
        const size_t own_b = plane_b * (size_t)(S.end - S.begin);
        if (o->terminal) MULTI_TRY(hipMemcpy(J0 + plane_b * S.hlo, (const char *)o->terminal + plane_b * S.begin, own_b, hipMemcpyHostToDevice));
        else MULTI_TRY(hipMemset(J0, 0, plane_b * (size_t)(S.end - S.begin + S.hlo + S.hhi)));
        MULTI_TRY(hipDeviceSynchronize());
    }
    auto stage_part = [&](hjb_multi_s::Slab &S, int k, int cur, hipStream_t stream) -> int {
        Handle *h = k < 0 ? S.whole : S.part[k];
        const int64_t row0 = k < 0 ? 0 : S.part_row0[k], own0 = k < 0 ? 0 : S.part_own0[k];
        const char *in = (const char *)S.whole->dJ[cur] + plane_b * row0;
        char *outp = (char *)S.whole->dJ[cur ^ 1] + plane_b * row0;
        const int st = launch_stage(h, in, outp, S.whole->d_idx + (size_t)(inner * own0) * m->isz, stream);
        if (st) m->err = h->err;
        return st;
    };
    const auto t0 = std::chrono::steady_clock::now();
    int cur = 0, done = 0, early = 0;
    double fprev = 0, iprev = 0, e = 0, e2 = 0;
    for (int k_s = o->n_stages; k_s >= 1; --k_s, ++done) {
        const int par = done & 1, ppar = par ^ 1;
        // ---- phase A: halo copies of J_{k+1} (buffer `cur`) on the copy streams ------------------------------------
        for (int i = 0; i < n; ++i) {
            auto &S = m->slabs[(size_t)i];
            if (!S.hlo && !S.hhi) continue;
            MULTI_TRY(hipSetDevice(S.device));
            if (done > 0) {           // the data: the neighbours' previous-stage output; the target: halo planes my own previous stage read
                MULTI_TRY(hipStreamWaitEvent(S.sx, S.done[ppar], 0));
                if (i > 0) MULTI_TRY(hipStreamWaitEvent(S.sx, m->slabs[(size_t)i - 1].done[ppar], 0));
                if (i + 1 < n) MULTI_TRY(hipStreamWaitEvent(S.sx, m->slabs[(size_t)i + 1].done[ppar], 0));
            }
            char *mine = (char *)S.whole->dJ[cur];
            if (S.hlo) {
                const auto &L = m->slabs[(size_t)i - 1];
                const char *src = (const char *)L.whole->dJ[cur] + plane_b * (size_t)(L.hlo + (L.end - L.begin) - S.hlo);
                MULTI_TRY(hipMemcpyPeerAsync(mine, S.device, src, L.device, plane_b * (size_t)S.hlo, S.sx));
            }
            if (S.hhi) {
                const auto &R = m->slabs[(size_t)i + 1];
                const char *src = (const char *)R.whole->dJ[cur] + plane_b * (size_t)R.hlo;
                MULTI_TRY(hipMemcpyPeerAsync(mine + plane_b * (size_t)(S.hlo + S.end - S.begin), S.device, src, R.device, plane_b * (size_t)S.hhi, S.sx));
            }
            MULTI_TRY(hipEventRecord(S.halo[par], S.sx));
        }
        // ---- phase B: interior, then (halos landed) the strips, on the compute streams --------------------------------
        for (int i = 0; i < n; ++i) {
            auto &S = m->slabs[(size_t)i];
            MULTI_TRY(hipSetDevice(S.device));
            if (done > 0) {           // the neighbours read buffer cur^1 (my output now) as their halo source one stage ago
                if (i > 0 && m->slabs[(size_t)i - 1].hhi) MULTI_TRY(hipStreamWaitEvent(S.sc, m->slabs[(size_t)i - 1].halo[ppar], 0));
                if (i + 1 < n && m->slabs[(size_t)i + 1].hlo) MULTI_TRY(hipStreamWaitEvent(S.sc, m->slabs[(size_t)i + 1].halo[ppar], 0));
            }
            int st = HJB_OK;
            if (S.part[0]) {
                // the strips on streams of their own, beside the interior: each launch of the column-sweep kernel lasts at
                // least one column (~0.2 ms), in line behind the interior two strips would cost more than the copies hide.
                // A strip stream waits for what the compute stream has waited for so far (event `fork`), and for the halos.
                MULTI_TRY(hipEventRecord(S.fork, S.sc));          // fork point: everything this stage depends on, before the interior
                for (int k = 1; k <= 2 && !st; ++k)
                    if (S.part[k]) MULTI_TRY(hipStreamWaitEvent(S.ss[k - 1], S.fork, 0));
                st = stage_part(S, 0, cur, S.sc);
                for (int k = 1; k <= 2 && !st; ++k)
                    if (S.part[k]) {
                        if (S.hlo || S.hhi) MULTI_TRY(hipStreamWaitEvent(S.ss[k - 1], S.halo[par], 0));
                        st = stage_part(S, k, cur, S.ss[k - 1]);
                        if (!st) {
                            MULTI_TRY(hipEventRecord(S.sdone[k - 1], S.ss[k - 1]));
                            MULTI_TRY(hipStreamWaitEvent(S.sc, S.sdone[k - 1], 0));
                        }
                    }
            } else {
                if (S.hlo || S.hhi) MULTI_TRY(hipStreamWaitEvent(S.sc, S.halo[par], 0));
                st = stage_part(S, -1, cur, S.sc);
            }
            if (st) return mfail(m, st, "stage launch on slab %d: %s", i, m->err.c_str());
            MULTI_TRY(hipEventRecord(S.done[par], S.sc));
        }
        // per-stage planes (Dynamic_Solver.m:100,105): plane k_s - 1 of the host arrays, each slab's states.  Issued once
        // EVERY slab's stage is enqueued: a copy into pageable host memory holds the host until that slab's stage has
        // finished, and the other slabs must be computing meanwhile (the output buffer is rewritten two stages on, idx one)
        if (o->J_stages || o->idx_stages)
            for (int i = 0; i < n; ++i) {
                auto &S = m->slabs[(size_t)i];
                MULTI_TRY(hipSetDevice(S.device));
                const size_t own = (size_t)(S.end - S.begin);
                const size_t at = (size_t)(k_s - 1) * (size_t)inner * (size_t)m->nl + (size_t)inner * (size_t)S.begin;
                if (o->J_stages)
                    MULTI_TRY(hipMemcpyAsync((char *)o->J_stages + at * esz, (const char *)S.whole->dJ[cur ^ 1] + plane_b * S.hlo, plane_b * own,
                                             hipMemcpyDeviceToHost, S.sc));
                if (o->idx_stages)
                    MULTI_TRY(hipMemcpyAsync((char *)o->idx_stages + at * m->isz, S.whole->d_idx, (size_t)inner * own * m->isz, hipMemcpyDeviceToHost, S.sc));
            }
        if (every_stage && !(o->monitor_period > 0 && (k_s % o->monitor_period) == 0)) {    // Dynamic_Solver.m:101: one line per stage
            for (auto &S : m->slabs) {
                MULTI_TRY(hipSetDevice(S.device));
                MULTI_TRY(hipStreamSynchronize(S.sc));
            }
            o->progress(o->progress_user, k_s, 0.0, 0.0, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        cur ^= 1;
        // ---- the early-stop monitor (Solver_pos_att.m:273-285): per-slab sums, added on the host ----------------------
        if (o->monitor_period > 0 && (k_s % o->monitor_period) == 0) {
            double sj = 0, si = 0;
            for (auto &S : m->slabs) {
                MULTI_TRY(hipSetDevice(S.device));
                const char *Jown = (const char *)S.whole->dJ[cur] + plane_b * S.hlo;
                if (launch_monitor_sums(m->dtype, false, Jown, S.whole->d_idx, (int32_t)m->isz, inner * (S.end - S.begin), S.whole->d_partials, S.whole->d_sums, S.sc) != HJB_OK)
                    return mfail(m, HJB_E_DEVICE, "monitor reduction launch failed");
            }
            for (auto &S : m->slabs) {
                double sums[2];
                MULTI_TRY(hipSetDevice(S.device));
                MULTI_TRY(hipMemcpyAsync(sums, S.whole->d_sums, sizeof sums, hipMemcpyDeviceToHost, S.sc));
                MULTI_TRY(hipStreamSynchronize(S.sc));
                sj += sums[0];
                si += sums[1];
            }
            e = sj - fprev; e2 = si - iprev; fprev = sj; iprev = si;
            if (o->progress) o->progress(o->progress_user, k_s, e, e2, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            if (std::fabs(e) < o->monitor_tol) { early = 1; ++done; break; }
        }
    }
    for (auto &S : m->slabs) {
        MULTI_TRY(hipSetDevice(S.device));
        MULTI_TRY(hipStreamSynchronize(S.sc));
        MULTI_TRY(hipStreamSynchronize(S.sx));
        MULTI_TRY(hipStreamSynchronize(S.ss[0]));
        MULTI_TRY(hipStreamSynchronize(S.ss[1]));
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (auto &S : m->slabs) {
        Handle *hs[4] = {S.whole, S.part[0], S.part[1], S.part[2]};
        MULTI_TRY(hipSetDevice(S.device));
        for (Handle *h : hs)
            if (h) {
                const int st = check_status(h, S.sc);
                if (st) return mfail(m, st, "%s", h->err.c_str());
            }
        const size_t own = (size_t)(S.end - S.begin);
        if (o->J_final) MULTI_TRY(hipMemcpy((char *)o->J_final + plane_b * S.begin, (const char *)S.whole->dJ[cur] + plane_b * S.hlo, plane_b * own, hipMemcpyDeviceToHost));
        if (o->idx_final) MULTI_TRY(hipMemcpy((char *)o->idx_final + (size_t)(inner * S.begin) * m->isz, S.whole->d_idx, (size_t)inner * own * m->isz, hipMemcpyDeviceToHost));
    }
    if (res) {
        res->stages_done = done;
        res->stopped_early = early;
        res->sweep_ms = ms;
        res->last_e = e;
        res->last_e2 = e2;
    }
    return HJB_OK;
#undef MULTI_TRY
}

int32_t hjb_create_multi_from(hjb_builder b, int32_t n_dev, const int32_t *devices, hjb_multi *out) {
    if (!b || !out) return bfail(b, HJB_E_INVALID, "null argument");
    hjb_problem p;
    const int st0 = builder_bind(b, &p);
    if (st0) return st0;
    const int st = hjb_create_multi(&p, n_dev, devices, out);
    if (st) b->err = g_last_error;
    return st;
}

int32_t hjb_solve_multi_flat(hjb_multi m, int32_t n_stages, int32_t monitor_period, double monitor_tol, const void *terminal,
                             void *J_final, void *idx_final, int32_t *stages_done, int32_t *stopped_early, double *sweep_ms) {
    hjb_solve_opts o{};
    o.n_stages = n_stages;
    o.monitor_period = monitor_period;
    o.monitor_tol = monitor_tol;
    o.terminal = terminal;
    o.J_final = J_final;
    o.idx_final = idx_final;
    hjb_result r{};
    const int st = hjb_solve_multi(m, &o, &r);
    if (stages_done) *stages_done = r.stages_done;
    if (stopped_early) *stopped_early = r.stopped_early;
    if (sweep_ms) *sweep_ms = r.sweep_ms;
    return st;
}

// ---- one process per GPU: a rank's slab as interior + boundary strips ------------------------------------------------
// What hjb_solve_multi does per slab and stage, for a host that runs ONE PROCESS PER GPU and moves the halo planes itself
// (MPI, RCCL through torch.distributed: hjbdp/sharded.py, bench.py --gpus N).  The library partitions the last state axis
// exactly as hjb_create_multi does, creates this rank's slab handle and - when the slab has an interior - the interior
// and strip handles over the same buffers, and enqueues a whole stage (fork, interior, strips behind the halos, join) in
// ONE call: the per-stage host work of a rank is the exchange plus this call.
struct hjb_rank_s {
    int device = 0, rank = 0, world = 1, begin = 0, end = 0, hlo = 0, hhi = 0, need_lo = 0, need_hi = 0, nl = 0;
    Handle *whole = nullptr;
    Handle *part[3] = {nullptr, nullptr, nullptr};     // interior, low strip, high strip (null: no split)
    int64_t part_row0[3] = {0, 0, 0}, part_own0[3] = {0, 0, 0};
    hipStream_t ss[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, halo = nullptr, sdone[2] = {nullptr, nullptr};
    int64_t inner = 0;
    size_t esz = 4, isz = 4;
    std::string err;
    // RCCL transport (hjb_rank_comm_init): the communicator, the transfer stream, an event that orders it behind the
    // compute stream, the monitor's reduction scratch, and the loopback switch of the one-GPU transport test
    void *comm = nullptr;
    hipStream_t xfer = nullptr;
    hipEvent_t xready = nullptr;
    double *d_partials = nullptr, *d_sums = nullptr;
    bool loopback = false;
    int dtype = HJB_F32, up_needs = 0, dn_needs = 0;
    int64_t xfer_delay_ticks = 0;     // option "xfer_delay_us": a spin of that length behind every exchange (link-latency emulation)
    bool monitor_single = false;      // option "monitor_single": hjb_rank_sweep's monitor in single precision (see there)
};

static int rfail(hjb_rank r, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (r) r->err = buf;
    g_last_error = buf;
    return code;
}

const char *hjb_rank_last_error(hjb_rank r) { return r ? r->err.c_str() : g_last_error.c_str(); }
static void rank_comm_release(hjb_rank r);

int32_t hjb_rank_destroy(hjb_rank r) {
    if (!r) return HJB_OK;
    (void)hipSetDevice(r->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i) {
        if (r->sdone[i]) (void)hipEventDestroy(r->sdone[i]);
        if (r->ss[i]) (void)hipStreamDestroy(r->ss[i]);
    }
    if (r->fork) (void)hipEventDestroy(r->fork);
    if (r->halo) (void)hipEventDestroy(r->halo);
    rank_comm_release(r);
    for (int i = 0; i < 3; ++i) if (r->part[i]) (void)hjb_destroy((hjb_handle)r->part[i]);
    if (r->whole) (void)hjb_destroy((hjb_handle)r->whole);
    delete r;
    return HJB_OK;
}

int32_t hjb_rank_create(const hjb_problem *p, int32_t device, int32_t rank, int32_t world, int32_t overlap, hjb_rank *out) {
    if (!p || !out) return rfail(nullptr, HJB_E_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return rfail(nullptr, HJB_E_INVALID, "rank %d of %d", rank, world);
    if (p->slab_begin || p->slab_end || p->halo_lo || p->halo_hi) return rfail(nullptr, HJB_E_INVALID, "hjb_rank_create partitions the grid itself: pass the whole problem");
    if (p->D < 1 || p->D > HJB_MAX_D) return rfail(nullptr, HJB_E_UNSUPPORTED, "D=%d", p->D);
    const int nl = p->n[p->D - 1];
    if (world > nl) return rfail(nullptr, HJB_E_INVALID, "more ranks (%d) than planes of the last axis (%d)", world, nl);
    hjb_info pin{};                              // the halo the tables imply + the label width: host arithmetic only
    int st;
    {
        int ib = 4, hl = 0, hh = 0;
        int64_t ns = 0;
        st = analyse_problem(p, &ib, &ns, &hl, &hh);
        if (st) return st;
        pin.idx_bytes = ib; pin.n_states = ns; pin.halo_needed_lo = hl; pin.halo_needed_hi = hh;
    }
    hjb_rank r = new hjb_rank_s();
    r->device = device; r->rank = rank; r->world = world; r->nl = nl;
    r->need_lo = pin.halo_needed_lo; r->need_hi = pin.halo_needed_hi;
    r->esz = p->dtype == HJB_F16S ? 2 : (p->dtype == HJB_F32 ? 4 : 8);
    r->isz = (size_t)pin.idx_bytes;
    r->inner = pin.n_states / nl;
    const int base = nl / world, rem = nl % world;
    auto range = [&](int k, int *b, int *e) { *b = k * base + std::min(k, rem); *e = *b + base + (k < rem ? 1 : 0); };
    range(rank, &r->begin, &r->end);
    r->hlo = std::min(r->need_lo, r->begin);
    r->hhi = std::min(r->need_hi, nl - r->end);
    r->dtype = p->dtype;
    r->up_needs = rank + 1 < world ? std::min(r->need_lo, r->end) : 0;        // my top planes -> rank + 1's lower halo
    r->dn_needs = rank > 0 ? std::min(r->need_hi, nl - r->begin) : 0;         // my bottom planes -> rank - 1's upper halo
    for (int k = 0; k < world; ++k) {            // a halo must come from the immediate neighbour only
        int b, e;
        range(k, &b, &e);
        int pb = 0, pe = 0, nb = 0, ne = 0;
        if (k > 0) range(k - 1, &pb, &pe);
        if (k + 1 < world) range(k + 1, &nb, &ne);
        if ((k > 0 && std::min(r->need_lo, b) > pe - pb) || (k + 1 < world && std::min(r->need_hi, nl - e) > ne - nb)) {
            (void)hjb_rank_destroy(r);
            return rfail(nullptr, HJB_E_INVALID, "halo (%d/%d planes) wider than a neighbouring slab: use fewer ranks or relabel the "
                         "state axes so that the last axis moves less", pin.halo_needed_lo, pin.halo_needed_hi);
        }
    }
    auto make = [&](int sb, int se, int hl, int hh, Handle **hout) {
        hjb_problem q = *p;
        if (world > 1) { q.slab_begin = sb; q.slab_end = se; q.halo_lo = hl; q.halo_hi = hh; }
        hjb_handle h = nullptr;
        const int s2 = hjb_create(&q, device, &h);
        *hout = (Handle *)h;
        return s2;
    };
    st = make(r->begin, r->end, r->hlo, r->hhi, &r->whole);
    const int lo_w = r->hlo ? r->need_lo : 0, hi_w = r->hhi ? r->need_hi : 0, owned = r->end - r->begin;
    if (!st && overlap && world > 1 && owned - lo_w - hi_w >= 1 && (lo_w || hi_w)) {
        const int view0 = r->begin - r->hlo;
        auto sub = [&](int k, int sb, int se, int hl, int hh) {
            r->part_row0[k] = (sb - hl) - view0;
            r->part_own0[k] = sb - r->begin;
            return make(sb, se, hl, hh, &r->part[k]);
        };
        st = sub(0, r->begin + lo_w, r->end - hi_w, std::min(r->need_lo, lo_w), std::min(r->need_hi, hi_w));
        if (!st && lo_w) st = sub(1, r->begin, r->begin + lo_w, r->hlo, std::min(r->need_hi, r->end - (r->begin + lo_w)));
        if (!st && hi_w) st = sub(2, r->end - hi_w, r->end, std::min(r->need_lo, (r->end - hi_w) - r->begin), r->hhi);
    }
    if (!st) {
        bool ok = hipSetDevice(device) == hipSuccess && hipEventCreateWithFlags(&r->fork, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&r->halo, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < 2 && ok; ++k)
            ok = hipEventCreateWithFlags(&r->sdone[k], hipEventDisableTiming) == hipSuccess &&
                 hipStreamCreateWithFlags(&r->ss[k], hipStreamNonBlocking) == hipSuccess;
        if (!ok) st = rfail(nullptr, HJB_E_DEVICE, "stream / event creation failed on device %d", device);
    }
    if (st) {
        const std::string keep = g_last_error;
        (void)hjb_rank_destroy(r);
        g_last_error = keep;
        return st;
    }
    *out = r;
    return HJB_OK;
}

// hjb_rank_create for hosts that describe the problem with the flat builder (MATLAB's calllib: one worker per GPU)
int32_t hjb_rank_create_from(hjb_builder b, int32_t device, int32_t rank, int32_t world, int32_t overlap, hjb_rank *out) {
    if (!b || !out) return bfail(b, HJB_E_INVALID, "null argument");
    hjb_problem p;
    const int st0 = builder_bind(b, &p);
    if (st0) return st0;
    const int st = hjb_rank_create(&p, device, rank, world, overlap, out);
    if (st) b->err = g_last_error;
    return st;
}

int32_t hjb_rank_info(hjb_rank r, int32_t *out10) {
    if (!r || !out10) return rfail(r, HJB_E_INVALID, "null argument");
    out10[0] = r->begin; out10[1] = r->end; out10[2] = r->hlo; out10[3] = r->hhi;
    out10[4] = r->part[0] ? 1 : 0;
    out10[5] = (r->part[0] ? r->part[0] : r->whole)->variant;
    out10[6] = r->need_lo; out10[7] = r->need_hi;
    out10[8] = (int32_t)r->isz; out10[9] = r->nl;
    return HJB_OK;
}

int32_t hjb_rank_set_option(hjb_rank r, const char *key, int64_t value) {
    if (!r || !key) return rfail(r, HJB_E_INVALID, "null argument");
    if (!strcmp(key, "xfer_delay_us")) {      // emulation only: every halo exchange takes this much longer (tools/emulate_ranks.py)
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, r->device) != hipSuccess || khz <= 0) khz = 100000;
        r->xfer_delay_ticks = value > 0 ? value * (int64_t)khz / 1000 : 0;
        return HJB_OK;
    }
    if (!strcmp(key, "comm_loopback")) {      // before hjb_rank_comm_init: the one-GPU transport test (hjbdp.h)
        if (r->comm) return rfail(r, HJB_E_INVALID, "comm_loopback must be set before hjb_rank_comm_init");
        r->loopback = value != 0;
        if (r->loopback) {                    // this rank plays both neighbours: it needs what it would have received
            r->up_needs = r->hlo;
            r->dn_needs = r->hhi;
        }
        return HJB_OK;
    }
    if (!strcmp(key, "monitor_single")) r->monitor_single = value != 0;      // ... and on to the handles (hjb_rank_get_option reads it there)
    Handle *hs[4] = {r->whole, r->part[0], r->part[1], r->part[2]};
    for (Handle *h : hs)
        if (h) {
            const int st = hjb_set_option((hjb_handle)h, key, value);
            if (st) return rfail(r, st, "%s", hjb_last_error((hjb_handle)h));
        }
    return HJB_OK;
}

int32_t hjb_rank_get_option(hjb_rank r, const char *key, int64_t *value) {
    if (!r) return rfail(r, HJB_E_INVALID, "null argument");
    return hjb_get_option((hjb_handle)(r->part[0] ? r->part[0] : r->whole), key, value);
}

int32_t hjb_rank_check_status(hjb_rank r, void *stream) {
    if (!r) return rfail(r, HJB_E_INVALID, "null argument");
    if (hipSetDevice(r->device) != hipSuccess) return rfail(r, HJB_E_DEVICE, "hipSetDevice failed");
    Handle *hs[4] = {r->whole, r->part[0], r->part[1], r->part[2]};
    for (Handle *h : hs)
        if (h) {
            const int st = check_status(h, (hipStream_t)stream);
            if (st) return rfail(r, st, "%s", h->err.c_str());
        }
    return HJB_OK;
}

int32_t hjb_rank_stage(hjb_rank r, const void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream, void *halo_stream) {
    if (!r || !dJ_in || !dJ_out) return rfail(r, HJB_E_INVALID, "null argument");
    hipStream_t cs = (hipStream_t)compute_stream, hs = (hipStream_t)halo_stream;
    const size_t plane_b = (size_t)r->inner * r->esz;
#define RANK_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return rfail(r, HJB_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
    RANK_TRY(hipSetDevice(r->device));
    auto stage_part = [&](int k, hipStream_t stream) -> int {
        Handle *h = k < 0 ? r->whole : r->part[k];
        const int64_t row0 = k < 0 ? 0 : r->part_row0[k], own0 = k < 0 ? 0 : r->part_own0[k];
        const int st = launch_stage(h, (const char *)dJ_in + plane_b * row0, (char *)dJ_out + plane_b * row0,
                                    d_idx ? (char *)d_idx + (size_t)(r->inner * own0) * r->isz : nullptr, stream);
        if (st) r->err = h->err;
        return st;
    };
    const bool halos = hs != nullptr && (r->hlo || r->hhi);
    if (!r->part[0]) {                       // no interior to overlap with: the halos first, then one kernel
        if (halos) {
            RANK_TRY(hipEventRecord(r->halo, hs));
            RANK_TRY(hipStreamWaitEvent(cs, r->halo, 0));
        }
        return stage_part(-1, cs);
    }
    // the strips run on streams of their own, beside the interior (see hjb_solve_multi): a strip stream waits for what the
    // compute stream holds so far (J_in complete) and for the halos; the compute stream joins them at the end
    RANK_TRY(hipEventRecord(r->fork, cs));
    for (int k = 1; k <= 2; ++k)
        if (r->part[k]) RANK_TRY(hipStreamWaitEvent(r->ss[k - 1], r->fork, 0));
    int st = stage_part(0, cs);
    if (st) return st;
    if (halos) RANK_TRY(hipEventRecord(r->halo, hs));
    for (int k = 1; k <= 2; ++k)
        if (r->part[k]) {
            if (halos) RANK_TRY(hipStreamWaitEvent(r->ss[k - 1], r->halo, 0));
            st = stage_part(k, r->ss[k - 1]);
            if (st) return st;
            RANK_TRY(hipEventRecord(r->sdone[k - 1], r->ss[k - 1]));
            RANK_TRY(hipStreamWaitEvent(cs, r->sdone[k - 1], 0));
        }
    return HJB_OK;
#undef RANK_TRY
}

// ---- RCCL inside the library: the halo exchange and the monitor's all-reduce of a rank, no torch, no MPI ------------------
// per stage `ncclGroupStart; ncclSend / ncclRecv x <= 4; ncclGroupEnd` on a transfer stream (one xGMI link per
// neighbour pair), every monitor period a 2-double ncclAllReduce.  librccl is dlopen'ed on first use: libhjbdp carries no link
// dependency on it (a single-GPU host never loads it; a process that already holds a librccl - torch's - shares it).
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, ncclUniqueIdBytes, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
constexpr int kNcclUint8 = 1, kNcclFloat64 = 8, kNcclSum = 0;      // rccl.h: ncclDataType_t / ncclRedOp_t

bool rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return true;
    const bool only_env = g_test_rccl_only_env.load() != 0;
    const char *names[] = {getenv("HJBDP_RCCL_LIB"), only_env ? nullptr : "librccl.so.1", only_env ? nullptr : "librccl.so",
                           only_env ? nullptr : "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) {
        if (!n || !n[0]) continue;
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) {
        const char *e = dlerror();        // ONE call: dlerror() clears the error state, a second call returns NULL
        g_rccl.why = std::string("dlopen(librccl.so.1): ") + (e ? e : "not found");
        return false;
    }
    auto sym = [&](const char *name) -> void * {
        void *f = dlsym(lib, name);
        if (!f) g_rccl.why = std::string("librccl lacks ") + name;
        return f;
    };
    RcclApi a;
    a.GetUniqueId = (int (*)(void *))sym("ncclGetUniqueId");
    a.CommInitRank = (int (*)(void **, int, ncclUniqueIdBytes, int))sym("ncclCommInitRank");
    a.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
    a.GroupStart = (int (*)())sym("ncclGroupStart");
    a.GroupEnd = (int (*)())sym("ncclGroupEnd");
    a.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))sym("ncclSend");
    a.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))sym("ncclRecv");
    a.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))sym("ncclAllReduce");
    a.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Send || !a.Recv || !a.AllReduce ||
        !a.GetErrorString) {
        dlclose(lib);
        return false;
    }
    a.lib = lib;
    a.why = g_rccl.why;
    g_rccl = a;
    return true;
}
}  // namespace

#define RCCL_TRY(r, expr)                                                                                     \
    do {                                                                                                      \
        const int e_ = (expr);                                                                                \
        if (e_ != 0) return rfail(r, HJB_E_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(e_));        \
    } while (0)
#define RANKH_TRY(r, expr)                                                                                    \
    do {                                                                                                      \
        const hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return rfail(r, HJB_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

// a fixed wall-clock delay on a stream (wall_clock64: the constant-rate counter): option "xfer_delay_us"
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static void rank_comm_release(hjb_rank r) {
    if (r->comm && g_rccl.lib) (void)g_rccl.CommDestroy(r->comm);
    r->comm = nullptr;
    if (r->xfer) (void)hipStreamDestroy(r->xfer);
    if (r->xready) (void)hipEventDestroy(r->xready);
    if (r->d_partials) (void)hipFree(r->d_partials);
    if (r->d_sums) (void)hipFree(r->d_sums);
    r->xfer = nullptr; r->xready = nullptr; r->d_partials = nullptr; r->d_sums = nullptr;
}

int32_t hjb_rank_comm_unique_id(void *id128_out) {
    if (!id128_out) return rfail(nullptr, HJB_E_INVALID, "null argument");
    if (!rccl_load()) return rfail(nullptr, HJB_E_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why.c_str());
    RCCL_TRY(nullptr, g_rccl.GetUniqueId(id128_out));
    return HJB_OK;
}

int32_t hjb_rank_comm_init(hjb_rank r, const void *id128) {
    if (!r || !id128) return rfail(r, HJB_E_INVALID, "null argument");
    if (r->comm) return rfail(r, HJB_E_INVALID, "this rank already has a communicator");
    if (!rccl_load()) return rfail(r, HJB_E_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why.c_str());
    RANKH_TRY(r, hipSetDevice(r->device));
    ncclUniqueIdBytes id;
    memcpy(id.internal, id128, sizeof id.internal);
    // loopback (option "comm_loopback", the one-GPU transport test): a communicator of ONE rank, both neighbours = this rank
    RCCL_TRY(r, g_rccl.CommInitRank(&r->comm, r->loopback ? 1 : r->world, id, r->loopback ? 0 : r->rank));
    hipError_t e = hipStreamCreateWithFlags(&r->xfer, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->xready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&r->d_partials, sizeof(double) * 2 * kReduceBlocks);
    if (e == hipSuccess) e = hipMalloc((void **)&r->d_sums, sizeof(double) * 2);
    if (e != hipSuccess) {          // leave nothing half-built: a retry must not be refused with "already has a communicator"
        rank_comm_release(r);
        return rfail(r, e == hipErrorOutOfMemory ? HJB_E_NOMEM : HJB_E_DEVICE, "hjb_rank_comm_init: %s", hipGetErrorString(e));
    }
    return HJB_OK;
}

// The halo exchange of dJ (this rank's haloed J buffer) on the library's transfer stream, ordered behind everything
// `compute_stream` holds at the time of the call (the stage that wrote dJ).  Returns at once; hjb_rank_stage's halo_stream
// argument = hjb_rank_transfer_stream(r) makes the boundary strips wait for it (hjb_rank_step does both).
int32_t hjb_rank_exchange(hjb_rank r, void *dJ, void *compute_stream) {
    if (!r || !dJ) return rfail(r, HJB_E_INVALID, "null argument");
    if (!r->comm) return rfail(r, HJB_E_INVALID, "hjb_rank_comm_init first");
    RANKH_TRY(r, hipSetDevice(r->device));
    RANKH_TRY(r, hipEventRecord(r->xready, (hipStream_t)compute_stream));
    RANKH_TRY(r, hipStreamWaitEvent(r->xfer, r->xready, 0));
    const size_t plane_b = (size_t)r->inner * r->esz;
    const int owned = r->end - r->begin;
    char *J = (char *)dJ;
    const int dn = r->loopback ? 0 : r->rank - 1, up = r->loopback ? 0 : r->rank + 1;
    if (!(r->dn_needs || r->hlo || r->up_needs || r->hhi)) return HJB_OK;
    RCCL_TRY(r, g_rccl.GroupStart());
    int e1 = 0;
    // towards rank - 1: my lowest owned planes are its upper halo; its top planes are my lower halo
    if (r->dn_needs && !e1) e1 = g_rccl.Send(J + plane_b * r->hlo, plane_b * r->dn_needs, kNcclUint8, dn, r->comm, r->xfer);
    if (r->up_needs && !e1) e1 = g_rccl.Send(J + plane_b * (r->hlo + owned - r->up_needs), plane_b * r->up_needs, kNcclUint8, up, r->comm, r->xfer);
    // loopback: what goes "down" comes back as my own upper halo, what goes "up" as my lower halo (receives posted in the
    // order the one peer's sends were)
    if (r->loopback) {
        if (r->hhi && !e1) e1 = g_rccl.Recv(J + plane_b * (r->hlo + owned), plane_b * r->hhi, kNcclUint8, 0, r->comm, r->xfer);
        if (r->hlo && !e1) e1 = g_rccl.Recv(J, plane_b * r->hlo, kNcclUint8, 0, r->comm, r->xfer);
    } else {
        if (r->hlo && !e1) e1 = g_rccl.Recv(J, plane_b * r->hlo, kNcclUint8, dn, r->comm, r->xfer);
        if (r->hhi && !e1) e1 = g_rccl.Recv(J + plane_b * (r->hlo + owned), plane_b * r->hhi, kNcclUint8, up, r->comm, r->xfer);
    }
    const int e2 = g_rccl.GroupEnd();
    if (e1 || e2) return rfail(r, HJB_E_DEVICE, "halo exchange: %s", g_rccl.GetErrorString(e1 ? e1 : e2));
    if (r->xfer_delay_ticks > 0) hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, r->xfer, (long long)r->xfer_delay_ticks);
    return HJB_OK;
}

void *hjb_rank_transfer_stream(hjb_rank r) { return r ? (void *)r->xfer : nullptr; }

// exchange + stage: one call per stage for a host that owns nothing but the two J buffers and the label buffer
int32_t hjb_rank_step(hjb_rank r, void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream) {
    if (!r) return rfail(r, HJB_E_INVALID, "null argument");
    if (r->world > 1 || r->loopback) {
        const int st = hjb_rank_exchange(r, dJ_in, compute_stream);
        if (st) return st;
    }
    return hjb_rank_stage(r, dJ_in, dJ_out, d_idx, compute_stream, (r->world > 1 || r->loopback) ? (void *)r->xfer : nullptr);
}

// The monitor's two sums (Solver_pos_att.m:274-275) over the WHOLE grid: this rank's owned planes reduced on the device
// (fixed tree, float64), then one 2-double ncclAllReduce; sums2 = {sum J, sum labels} on every rank.
int32_t hjb_rank_monitor_sums(hjb_rank r, const void *dJ, const void *d_idx, void *compute_stream, double *sums2) {
    if (!r || !dJ || !sums2) return rfail(r, HJB_E_INVALID, "null argument");
    if (!r->comm) return rfail(r, HJB_E_INVALID, "hjb_rank_comm_init first");
    RANKH_TRY(r, hipSetDevice(r->device));
    hipStream_t cs = (hipStream_t)compute_stream;
    const size_t plane_b = (size_t)r->inner * r->esz;
    const int64_t n = r->inner * (int64_t)(r->end - r->begin);
    // option "monitor_single" at world == 1: the library's stated float32 tree over the whole grid, exactly hjb_solve's sum.
    // Over several ranks a float32 running sum in one fixed order does not exist: each rank sums its planes in float64 (fixed
    // tree) and the all-reduce adds the ranks in ITS order - reproducible for a given world size, not bit-identical to
    // hjb_solve's sum; hjb_rank_sweep then forms the difference and the comparison in single (below).
    const bool single_tree = r->monitor_single && r->world == 1 && !r->loopback && r->dtype != HJB_F64;
    if (launch_monitor_sums(r->dtype, single_tree, (const char *)dJ + plane_b * r->hlo, d_idx, (int32_t)r->isz, n, r->d_partials, r->d_sums, cs) != HJB_OK)
        return rfail(r, HJB_E_DEVICE, "monitor reduction launch failed");
    if (!d_idx) RANKH_TRY(r, hipMemsetAsync(r->d_sums + 1, 0, sizeof(double), cs));
    RCCL_TRY(r, g_rccl.AllReduce(r->d_sums, r->d_sums, 2, kNcclFloat64, kNcclSum, r->comm, cs));
    RANKH_TRY(r, hipMemcpyAsync(sums2, r->d_sums, 2 * sizeof(double), hipMemcpyDeviceToHost, cs));
    RANKH_TRY(r, hipStreamSynchronize(cs));
    return HJB_OK;
}

// The whole backward sweep of a rank: terminal cost in dJ0 (haloed layout, owned planes filled), per stage exchange + stage
// ping-ponging dJ0 / dJ1, the monitor every `monitor_period` stages (the same stop decision on every rank: the sums are
// all-reduced).  *final_in_0 says which buffer holds the last stage.  What SURVEY 8b (iii)'s C++ driver (tools/bench_ranks.cpp)
// and a MATLAB worker per GPU call.
int32_t hjb_rank_sweep(hjb_rank r, int32_t n_stages, int32_t monitor_period, double monitor_tol, void *dJ0, void *dJ1, void *d_idx,
                       void *compute_stream, int32_t *stages_done, int32_t *stopped_early, int32_t *final_in_0, double *sweep_ms) {
    if (!r || !dJ0 || !dJ1 || n_stages < 0) return rfail(r, HJB_E_INVALID, "bad argument");
    if (r->world > 1 && !r->comm) return rfail(r, HJB_E_INVALID, "hjb_rank_comm_init first (world > 1)");
    if (monitor_period > 0 && !r->comm) return rfail(r, HJB_E_INVALID, "the monitor needs a communicator (hjb_rank_comm_init), also at world == 1");
    RANKH_TRY(r, hipSetDevice(r->device));
    hipStream_t cs = (hipStream_t)compute_stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    void *J[2] = {dJ0, dJ1};
    int cur = 0, done = 0, early = 0, st = HJB_OK;
    double fprev = 0.0;
    // Solver_pos_att.m:276-282 with a single fsum50: the difference and `abs(e) < tol` are single-precision (hjb_solve's rule)
    const bool msingle = r->monitor_single && r->dtype != HJB_F64;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventRecord(e0, cs) != hipSuccess)
        st = rfail(r, HJB_E_DEVICE, "sweep: event set-up failed: %s", hipGetErrorString(hipGetLastError()));
    for (int k_s = n_stages; k_s >= 1 && !st; --k_s) {
        st = hjb_rank_step(r, J[cur], J[1 - cur], d_idx, compute_stream);
        if (st) break;
        cur = 1 - cur;
        ++done;
        if (monitor_period > 0 && (k_s % monitor_period) == 0) {
            double sums[2];
            st = hjb_rank_monitor_sums(r, J[cur], d_idx, compute_stream, sums);
            if (st) break;
            const double e = msingle ? (double)((float)sums[0] - (float)fprev) : sums[0] - fprev;
            fprev = sums[0];
            if (msingle ? (std::fabs((float)e) < (float)monitor_tol) : (std::fabs(e) < monitor_tol)) { early = 1; break; }
        }
    }
    if (!st) {
        (void)hipEventRecord(e1, cs);
        if (hipEventSynchronize(e1) != hipSuccess) st = rfail(r, HJB_E_DEVICE, "sweep: synchronisation failed");
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (sweep_ms) *sweep_ms = ms;
        if (!st) st = hjb_rank_check_status(r, compute_stream);
    }
    if (e0) (void)hipEventDestroy(e0);      // one exit: the events never leak
    if (e1) (void)hipEventDestroy(e1);
    if (stages_done) *stages_done = done;
    if (stopped_early) *stopped_early = early;
    if (final_in_0) *final_in_0 = cur == 0;
    return st;
}

// ---- device-buffer helpers -------------------------------------------------------------------------------------------
// hjb_backup_stage_device runs on buffers the caller owns.  A host without a HIP binding of its own (MATLAB, plain C)
// gets them here: allocation, copies, free memory, a separable fill and a gather - enough to drive grids that never
// exist on the host (C3: 51^6 states, 70 GB per buffer).
int32_t hjb_device_malloc(int32_t device, int64_t bytes, void **out) {
    if (!out || bytes < 0) return fail(nullptr, HJB_E_INVALID, "hjb_device_malloc: bad argument");
    *out = nullptr;
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d) failed", device);
    void *d = nullptr;
    const hipError_t e = hipMalloc(&d, (size_t)std::max<int64_t>(bytes, 16));
    if (e != hipSuccess) return fail(nullptr, HJB_E_NOMEM, "hipMalloc of %lld bytes: %s", (long long)bytes, hipGetErrorString(e));
    *out = d;
    return HJB_OK;
}

int32_t hjb_device_free(int32_t device, void *p) {
    if (!p) return HJB_OK;
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d) failed", device);
    (void)hipDeviceSynchronize();
    return hipFree(p) == hipSuccess ? HJB_OK : fail(nullptr, HJB_E_DEVICE, "hipFree failed");
}

int32_t hjb_device_mem_info(int32_t device, int64_t *free_bytes, int64_t *total_bytes) {
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d) failed", device);
    size_t f = 0, t = 0;
    if (hipMemGetInfo(&f, &t) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipMemGetInfo failed");
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return HJB_OK;
}

int32_t hjb_device_copy(int32_t device, void *dst, const void *src, int64_t bytes, int32_t kind) {
    if (!dst || !src || bytes < 0) return fail(nullptr, HJB_E_INVALID, "hjb_device_copy: bad argument");
    const hipMemcpyKind k = kind == HJB_COPY_H2D ? hipMemcpyHostToDevice : kind == HJB_COPY_D2H ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (kind < HJB_COPY_H2D || kind > HJB_COPY_D2D) return fail(nullptr, HJB_E_INVALID, "hjb_device_copy: kind %d", kind);
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d) failed", device);
    const hipError_t e = hipMemcpy(dst, src, (size_t)bytes, k);
    return e == hipSuccess ? HJB_OK : fail(nullptr, HJB_E_DEVICE, "hipMemcpy: %s", hipGetErrorString(e));
}

int32_t hjb_device_fill_separable(hjb_handle hh, const void *const *vecs, void *dJ, void *stream) {
    Handle *h = (Handle *)hh;
    if (!h || !vecs || !dJ) return fail(h, HJB_E_INVALID, "null argument");
    if (h->j_elems != h->n_owned) return fail(h, HJB_E_UNSUPPORTED, "hjb_device_fill_separable fills whole grids");
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    HIP_TRY(h, hipSetDevice(h->device));
    const int D = h->hp.D;
    const size_t tsz = h->dtype == HJB_F64 ? 8 : 4;
    DSeparable S{};
    std::vector<void *> tmp;
    for (int a = 0; a < D; ++a) {
        if (!vecs[a]) return fail(h, HJB_E_INVALID, "vecs[%d] is null", a);
        void *d = nullptr;
        if (hipMalloc(&d, (size_t)h->prob.n[a] * tsz) != hipSuccess) { for (void *t : tmp) (void)hipFree(t); return fail(h, HJB_E_NOMEM, "fill vectors"); }
        tmp.push_back(d);
        if (hipMemcpy(d, vecs[a], (size_t)h->prob.n[a] * tsz, hipMemcpyHostToDevice) != hipSuccess) { for (void *t : tmp) (void)hipFree(t); return fail(h, HJB_E_DEVICE, "fill vectors"); }
        S.v[a] = d;
        S.n[a] = h->prob.n[a];
    }
    S.D = D;
    S.total = h->n_owned;
    const unsigned grid = (unsigned)std::min<int64_t>((h->n_owned + 255) / 256, 256 * 64);
    if (h->dtype == HJB_F16S) hipLaunchKernelGGL((k_fill_separable<float, _Float16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, S, (_Float16 *)dJ);
    else if (h->dtype == HJB_F32) hipLaunchKernelGGL((k_fill_separable<float, float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, S, (float *)dJ);
    else hipLaunchKernelGGL((k_fill_separable<double, double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, S, (double *)dJ);
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);      // the vectors are freed below
    for (void *t : tmp) (void)hipFree(t);
    if (e != hipSuccess) return fail(h, HJB_E_DEVICE, "hjb_device_fill_separable: %s", hipGetErrorString(e));
    return HJB_OK;
}

int32_t hjb_device_gather(int32_t device, const void *d_src, int32_t elem_bytes, const int64_t *sel, int64_t n_sel, void *out) {
    if (!d_src || !sel || !out || n_sel < 0) return fail(nullptr, HJB_E_INVALID, "hjb_device_gather: bad argument");
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return fail(nullptr, HJB_E_INVALID, "hjb_device_gather: elem_bytes %d", elem_bytes);
    if (n_sel == 0) return HJB_OK;
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d) failed", device);
    void *dsel = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc(&dsel, (size_t)n_sel * 8);
    if (e == hipSuccess) e = hipMalloc(&dout, (size_t)n_sel * elem_bytes);
    if (e == hipSuccess) e = hipMemcpy(dsel, sel, (size_t)n_sel * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gather_bytes, dim3((unsigned)std::min<int64_t>((n_sel + 255) / 256, 65536)), dim3(256), 0, nullptr,
                           (const unsigned char *)d_src, elem_bytes, (const int64_t *)dsel, n_sel, (unsigned char *)dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n_sel * elem_bytes, hipMemcpyDeviceToHost);
    if (dsel) (void)hipFree(dsel);
    if (dout) (void)hipFree(dout);
    return e == hipSuccess ? HJB_OK : fail(nullptr, HJB_E_DEVICE, "hjb_device_gather: %s", hipGetErrorString(e));
}

}  // extern "C"
