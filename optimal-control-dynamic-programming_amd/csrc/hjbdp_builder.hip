// hjbdp_builder.hip - the flat builder API of libhjbdp (MATLAB loadlibrary / calllib cannot marshal hjb_problem).
// gfx950 (MI355X) only; no CPU fallback - without a HIP device every compute entry point returns HJB_E_DEVICE.
#include "hjbdp_host.h"

using namespace hjbhost;

extern "C" {

// ---- flat builder API (MATLAB loadlibrary/calllib cannot marshal hjb_problem) ---------------------------------------

int bfail(hjb_builder b, int code, const char *fmt, ...) {
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
    int64_t nS = 1;
    for (int a = 0; a < D; ++a) {
        if (n[a] < 2) return bfail(nullptr, HJB_E_INVALID, "n[%d]=%d < 2", a, n[a]);
        if (nS > kMaxStates / n[a]) return bfail(nullptr, HJB_E_UNSUPPORTED, "more than 2^40 grid points (axes 0..%d)", a);
        nS *= n[a];
    }
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
    for (int i = 0; i < len; ++i)
        if (!std::isfinite(knots[i])) return bfail(b, HJB_E_INVALID, "knots of axis %d: element %d is not finite", axis, i);
    for (int i = 0; i + 1 < len; ++i)
        if (!(knots[i + 1] > knots[i])) return bfail(b, HJB_E_INVALID, "knots of axis %d not strictly increasing at %d", axis, i);
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
    {
        const int64_t bad = first_nonfinite(data, count, esz == 8);
        if (bad >= 0) return bfail(b, HJB_E_INVALID, "%s (mask 0x%x): element %lld is not finite", what, mask, (long long)bad);
    }
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
        const int64_t bad = first_nonfinite(t[i], (int64_t)ne, false);
        if (bad >= 0) return bfail(b, HJB_E_INVALID, "model table %d: element %lld is not finite", i, (long long)bad);
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

int builder_bind(hjb_builder b, hjb_problem *out) {      // the builder's problem with its pointers bound
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

}  // extern "C"
