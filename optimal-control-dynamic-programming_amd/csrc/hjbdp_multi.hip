// hjbdp_multi.hip - hjb_create_multi / hjb_solve_multi: one process, the grid partitioned over several GPUs.
// gfx950 (MI355X) only; no CPU fallback - without a HIP device every compute entry point returns HJB_E_DEVICE.
#include "hjbdp_host.h"

using namespace hjbhost;

extern "C" {

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

}  // extern "C"
