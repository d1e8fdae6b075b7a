// hjbdp_batch.hip - hjb_solve_batch: several independent sweeps of ONE kernel shape side by side, ONE launch per stage for all of them
// (the column-sweep kernel, kernels_colsweep.h, or the table kernel's 32-bit form, kernels_tabled.h).
// gfx950 (MI355X) only; no CPU fallback - without a HIP device every compute entry point returns HJB_E_DEVICE.
//
// Solver_pos_att.simplified_run (pos-att/Solver_pos_att.m:197-242) sweeps four independent channels - x, y, z and the thruster-failure
// variant of x - of 2.7e5 states each.  A stage kernel of one channel is a launch boundary plus one wave's chain of round trips
// (~10 - 14 us whatever the grid), and of four such chains on four streams the device runs two at full rate (profiles/r05_pos_att_run.log):
// the four sweeps took two rounds.  Here the channels are the y dimension of one launch of the column-sweep kernel (kernels_colsweep.h:
// k_backup_colsweep_batch), the stage loop is one chain, the early-stop monitor (:268-285) keeps its own sums, difference and stop
// decision per channel, and a stopped channel drops out of the launches that follow.  Results per channel are those of hjb_solve, bit
// for bit (the same kernel body on the same buffers).
#include "hjbdp_host.h"

using namespace hjbhost;

extern "C" {

int32_t hjb_solve_batch(int32_t n, const hjb_handle *hs, const hjb_solve_opts *const *opts, hjb_result *const *res) {
    if (n < 1 || !hs || !opts) return fail(nullptr, HJB_E_INVALID, "hjb_solve_batch: bad argument");
    if (n > kCsBatchMax) return fail(nullptr, HJB_E_UNSUPPORTED, "hjb_solve_batch: at most %d problems", kCsBatchMax);
    Handle *H[kCsBatchMax];
    for (int i = 0; i < n; ++i) {
        H[i] = (Handle *)hs[i];
        if (!H[i] || !opts[i]) return fail(nullptr, HJB_E_INVALID, "hjb_solve_batch: problem %d is null", i);
    }
    // ---- what can run as one launch: one kernel shape and one loop --------------------------------------------------------------------
    //   the column-sweep kernel (variant 7) in its usual form, one group axis and cost typing: Solver_pos_att's channels;
    //   the table kernel (variant 5) in its 32-bit form, one (dtype, D <= 4): Solver_attitude.simplified_run's channels, Solver_pos_att's
    //   channels in the reference's own axis order
    const hjb_solve_opts &o0 = *opts[0];
    if (o0.n_stages < 1) return fail(H[0], HJB_E_INVALID, "n_stages=%d", o0.n_stages);
    const bool tabled = H[0]->variant == 5;
    int ng = 0;
    for (int i = 0; i < n; ++i) {
        Handle *h = H[i];
        const hjb_solve_opts &o = *opts[i];
        if (tabled) {
            const bool i32 = h->tabled_i32 && h->tabled_i32_on && (int64_t)h->grid * h->block <= kTab32MaxThreads;
            if (h->variant != 5 || !h->dtb || !i32 || h->j_elems != h->n_owned || h->hp.D > 4)
                return fail(h, HJB_E_UNSUPPORTED, "hjb_solve_batch: problem %d does not run on the table kernel's 32-bit form (variant %d); "
                            "sweep the problems side by side with hjb_solve on threads of their own", i, h->variant);
            if (h->device != H[0]->device || h->dtype != H[0]->dtype || h->hp.D != H[0]->hp.D || h->block != H[0]->block)
                return fail(h, HJB_E_UNSUPPORTED, "hjb_solve_batch: problem %d has another device, dtype, dimension or block size than problem 0", i);
        } else {
            const bool fastcost = h->hcs.ncu == 1 && h->hp.n_cost_prefix > 0;
            if (h->variant != 7 || !h->dcs || !h->dtb || h->dtype != HJB_F32 || h->j_elems != h->n_owned || !h->hcs.dpp || !fastcost ||
                (h->hcs.coop && h->cc_grid > 0))
                return fail(h, HJB_E_UNSUPPORTED, "hjb_solve_batch: problem %d does not run on the column-sweep kernel in its usual form (variant %d); "
                            "sweep the problems side by side with hjb_solve on threads of their own", i, h->variant);
            if (h->device != H[0]->device || h->hcs.gax != H[0]->hcs.gax || h->cost64 != H[0]->cost64)
                return fail(h, HJB_E_UNSUPPORTED, "hjb_solve_batch: problem %d has another device, group axis or cost typing than problem 0", i);
            ng = std::max(ng, (int)h->hcs.ng);
        }
        if (o.n_stages != o0.n_stages || o.monitor_period != o0.monitor_period)
            return fail(h, HJB_E_UNSUPPORTED, "hjb_solve_batch: one stage count and one monitor period for all problems");
        if (o.J_stages || o.idx_stages || o.probe || (o.progress && o.progress_every_stage))
            return fail(h, HJB_E_UNSUPPORTED, "hjb_solve_batch: no per-stage outputs, probe or per-stage progress (use hjb_solve)");
    }
    Handle *h0 = H[0];
    HIP_TRY(h0, hipSetDevice(h0->device));
    std::shared_lock<std::shared_mutex> unsafe_lk(g_capture_mu);
    for (int i = 0; i < n; ++i) {
        const int st = ensure_work(H[i]);
        if (st) return st;
    }
    // Parts per column.  A handle's own choice (colsweep_split) is made for a launch that is ALONE on the device: as many parts as
    // fill the wave slots, because such a launch is one wave's chain of round trips.  Every part primes its rows again, so that
    // choice does up to twice the work per column - the right price for latency, the wrong one for a batch, whose launches hold n
    // problems' columns: here the parts are what lets ALL the batch's columns fill about one round of the wave slots
    // (profiles/r06_batch_split.log: the reference's channels, 4 x 450 columns of 20 steps, 10 parts each when alone - 2 to 4 here).
    int old_split[kCsBatchMax];
    bool resplit = false;
    for (int i = 0; i < n; ++i) old_split[i] = H[i]->cs_split;
    auto restore_split = [&]() {              // every handle gets its own parts back, on every way out
        if (!resplit) return;
        for (int i = 0; i < n; ++i) {
            Handle *h = H[i];
            if (h->cs_split == old_split[i]) continue;
            h->cs_split = old_split[i];
            colsweep_split(h);
            (void)colsweep_upload(h);
            choose_launch(h);
        }
    };
    if (!tabled) {
        int64_t columns = 0;
        for (int i = 0; i < n; ++i) {
            const DParams &P = H[i]->hp;
            columns += (int64_t)((P.n[0] + kCsDppLanes - 1) / kCsDppLanes) * P.n[2] * P.n[3];
        }
        const int64_t slots = (int64_t)(h0->cost64 ? 5 : 6) * 4 * 256;                   // waves per SIMD of the form that runs (82 / 80 registers) x SIMDs
        const int s_batch = (int)std::max<int64_t>(1, slots / std::max<int64_t>(columns, 1));
        for (int i = 0; i < n; ++i) {
            Handle *h = H[i];
            if (h->cs_split == 0 && s_batch < (int)h->hcs.split) {      // (an explicit option "cs_split" stands)
                h->cs_split = s_batch;
                resplit = true;
                colsweep_split(h);
                const int ust = colsweep_upload(h);
                if (ust) { restore_split(); return ust; }
                choose_launch(h);
            }
        }
    }
    if (!h0->stream) {
        const hipError_t se = hipStreamCreateWithFlags(&h0->stream, hipStreamNonBlocking);
        if (se != hipSuccess) { restore_split(); return fail(h0, HJB_E_DEVICE, "hipStreamCreateWithFlags failed: %s", hipGetErrorString(se)); }
    }
    hipStream_t stream = h0->stream;
    DCsBatch hb;
    memset(&hb, 0, sizeof hb);
    unsigned gmax = 1;
    for (int i = 0; i < n; ++i) {
        Handle *h = H[i];
        hb.P[i] = h->dp; hb.TB[i] = h->dtb; hb.CS[i] = h->dcs;
        hb.J[i][0] = h->dJ[0]; hb.J[i][1] = h->dJ[1];
        hb.idx[i] = h->d_idx;
        hb.grid[i] = (uint32_t)h->grid;
        gmax = std::max(gmax, (unsigned)h->grid);
        const size_t jb = (size_t)h->n_owned * h->esz;
        const hipError_t te = opts[i]->terminal ? hipMemcpy(h->dJ[0], opts[i]->terminal, jb, hipMemcpyHostToDevice) : hipMemset(h->dJ[0], 0, jb);
        if (te != hipSuccess) { restore_split(); return fail(h, HJB_E_DEVICE, "terminal cost of problem %d: %s", i, hipGetErrorString(te)); }
    }
    DCsBatch *dB = nullptr;
    hipGraphExec_t gexec = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto cleanup = [&]() {
        std::shared_lock<std::shared_mutex> lk(g_capture_mu, std::defer_lock);
        if (!unsafe_lk.owns_lock()) lk.lock();
        if (gexec) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; }
        if (ev0) { (void)hipEventDestroy(ev0); ev0 = nullptr; }
        if (ev1) { (void)hipEventDestroy(ev1); ev1 = nullptr; }
        if (dB) { (void)hipFree(dB); dB = nullptr; }
        (void)hipStreamSynchronize(stream);       // (nothing of this batch is in flight when the handles get their own parts back)
        restore_split();
    };
#define BATCH_TRY(expr)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            cleanup();                                                                             \
            return fail(h0, HJB_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));          \
        }                                                                                          \
    } while (0)
    BATCH_TRY(hipMalloc((void **)&dB, sizeof hb));
    BATCH_TRY(hipMemcpy(dB, &hb, sizeof hb, hipMemcpyHostToDevice));
    BATCH_TRY(sync_setup());
    StageArgs a;
    a.grid = gmax;
    a.block = tabled ? (unsigned)h0->block : 256u;
    a.st = stream;
    a.dtype = tabled ? h0->dtype : HJB_F32;
    a.D = h0->hp.D;
    const int gax = h0->hcs.gax;
    const bool c64 = h0->cost64;
    auto launch = [&](uint32_t mask, int parity) -> int {
        if (tabled) {
            if (stage_tabled_batch(a, n, hb, mask, parity)) return fail(h0, HJB_E_DEVICE, "hjb_solve_batch: no batched instantiation (D = %d)", a.D);
            return HJB_OK;
        }
        if (stage_colsweep_batch(a, n, dB, mask, parity, gax, ng, c64)) return fail(h0, HJB_E_DEVICE, "hjb_solve_batch: no batched instantiation (%d groups)", ng);
        return HJB_OK;
    };
    uint32_t mask = n >= 32 ? 0xffffffffu : ((1u << n) - 1u), gmask = 0;
    const bool want_graph = h0->use_graph && o0.n_stages >= 2 * kGraphStages;
    unsafe_lk.unlock();
    // kGraphStages ping-pong launches for the problems of `mask`, starting and ending in buffer 0; re-captured when a problem stops
    auto capture = [&]() -> int {
        std::unique_lock<std::shared_mutex> capture_lk(g_capture_mu);
        if (gexec) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; }
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return fail(h0, HJB_E_DEVICE, "hjb_solve_batch: stream capture failed");
        int cst = HJB_OK;
        for (int i = 0; i < kGraphStages / 2 && cst == HJB_OK; ++i) {
            cst = launch(mask, 0);
            if (cst == HJB_OK) cst = launch(mask, 1);
        }
        hipError_t ce = hipStreamEndCapture(stream, &graph);
        if (cst != HJB_OK || ce != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            return fail(h0, HJB_E_DEVICE, "hjb_solve_batch: stage-loop graph capture failed: %s", hipGetErrorString(ce));
        }
        ce = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ce != hipSuccess) return fail(h0, HJB_E_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(ce));
        gmask = mask;
        return HJB_OK;
    };
    BATCH_TRY(hipEventCreate(&ev0));
    BATCH_TRY(hipEventCreate(&ev1));
    BATCH_TRY(hipEventRecord(ev0, stream));
    int parity = 0;                         // the buffer that holds the current cost-to-go of every problem still running
    int done[kCsBatchMax] = {0}, early[kCsBatchMax] = {0}, final_parity[kCsBatchMax] = {0};
    double fprev[kCsBatchMax] = {0}, iprev[kCsBatchMax] = {0}, e[kCsBatchMax] = {0}, e2[kCsBatchMax] = {0};
    int st = HJB_OK;
    int k_s = o0.n_stages;
    auto count = [&](int stages) { for (int i = 0; i < n; ++i) if ((mask >> i) & 1u) done[i] += stages; };
    while (k_s >= 1 && mask) {
        int stop = 1;
        if (o0.monitor_period > 0) stop = std::max(1, (k_s / o0.monitor_period) * o0.monitor_period);
        int run = k_s - stop + 1;
        if (want_graph && run >= kGraphStages) {
            if (parity == 1) {               // a replay starts in buffer 0: one eager stage
                st = launch(mask, parity);
                if (st) { cleanup(); return st; }
                parity ^= 1; count(1); --run; --k_s;
            }
            if (run >= kGraphStages && (!gexec || gmask != mask)) {
                st = capture();
                if (st) { cleanup(); return st; }
            }
            while (run >= kGraphStages) {
                BATCH_TRY(hipGraphLaunch(gexec, stream));
                count(kGraphStages); run -= kGraphStages; k_s -= kGraphStages;
            }
        }
        for (; run > 0; --run, --k_s) {
            st = launch(mask, parity);
            if (st) { cleanup(); return st; }
            parity ^= 1;
            count(1);
        }
        // here k_s == stop - 1: the stage just computed has reference index `stop` (Solver_pos_att.m:273-285, per problem)
        if (o0.monitor_period > 0 && (stop % o0.monitor_period) == 0) {
            double sums[kCsBatchMax][2];
            for (int i = 0; i < n; ++i) {
                if (!((mask >> i) & 1u)) continue;
                Handle *h = H[i];
                if (launch_monitor_sums(h->dtype, opts[i]->monitor_single != 0 || h->monitor_single, h->dJ[parity], h->d_idx, h->idx_bytes, h->n_owned,
                                        h->d_partials, h->d_sums, stream) != HJB_OK) { cleanup(); return fail(h, HJB_E_DEVICE, "monitor reduction launch failed"); }
            }
            unsafe_lk.lock();
            for (int i = 0; i < n; ++i)
                if ((mask >> i) & 1u) BATCH_TRY(hipMemcpyAsync(sums[i], H[i]->d_sums, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
            BATCH_TRY(hipStreamSynchronize(stream));
            float ms = 0;
            bool timed = false;
            unsafe_lk.unlock();
            for (int i = 0; i < n; ++i) {
                if (!((mask >> i) & 1u)) continue;
                Handle *h = H[i];
                const hjb_solve_opts &o = *opts[i];
                const bool msingle = (o.monitor_single != 0 || h->monitor_single) && h->dtype != HJB_F64;
                e[i] = msingle ? (double)((float)sums[i][0] - (float)fprev[i]) : sums[i][0] - fprev[i];
                e2[i] = sums[i][1] - iprev[i];
                fprev[i] = sums[i][0];
                iprev[i] = sums[i][1];
                if (o.progress) {
                    if (!timed) {
                        (void)hipEventRecord(ev1, stream);
                        (void)hipEventSynchronize(ev1);
                        (void)hipEventElapsedTime(&ms, ev0, ev1);
                        timed = true;
                    }
                    o.progress(o.progress_user, stop, e[i], e2[i], ms * 1e-3);
                }
                if (msingle ? (std::fabs((float)e[i]) < (float)o.monitor_tol) : (std::fabs(e[i]) < o.monitor_tol)) {
                    early[i] = 1;
                    final_parity[i] = parity;
                    mask &= ~(1u << i);           // this problem's sweep ends here: its result stays in the buffer it was just written to
                }
            }
        }
    }
    for (int i = 0; i < n; ++i)
        if (!early[i]) final_parity[i] = parity;
    BATCH_TRY(hipEventRecord(ev1, stream));
    BATCH_TRY(hipEventSynchronize(ev1));
    float ms = 0;
    BATCH_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    unsafe_lk.lock();
    for (int i = 0; i < n; ++i) {
        Handle *h = H[i];
        st = check_status(h, stream);
        if (st) { cleanup(); return st; }
        const size_t jb = (size_t)h->n_owned * h->esz, ib = (size_t)h->n_owned * h->idx_bytes;
        if (opts[i]->J_final) BATCH_TRY(hipMemcpy(opts[i]->J_final, h->dJ[final_parity[i]], jb, hipMemcpyDeviceToHost));
        if (opts[i]->idx_final) BATCH_TRY(hipMemcpy(opts[i]->idx_final, h->d_idx, ib, hipMemcpyDeviceToHost));
        if (res && res[i]) {
            res[i]->stages_done = done[i];
            res[i]->stopped_early = early[i];
            res[i]->sweep_ms = ms;
            res[i]->last_e = e[i];
            res[i]->last_e2 = e2[i];
        }
    }
    cleanup();
    return HJB_OK;
#undef BATCH_TRY
}

}  // extern "C"
