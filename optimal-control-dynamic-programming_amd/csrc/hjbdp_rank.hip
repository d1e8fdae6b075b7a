// hjbdp_rank.hip - hjb_rank_*: one process per GPU, and the RCCL transport inside the library.
// gfx950 (MI355X) only; no CPU fallback - without a HIP device every compute entry point returns HJB_E_DEVICE.
#include <dlfcn.h>
// RCCL's own declarations: types, enums and - through decltype - the signatures of every entry point this unit calls.  The
// library is still dlopen'ed on first use (no link dependency); what the header gives is that an enum value, the size of
// ncclUniqueId or an argument list can no longer drift from the librccl the image ships without this unit failing to build.
#include <rccl/rccl.h>

#include "hjbdp_host.h"

using namespace hjbhost;

static_assert(NCCL_UNIQUE_ID_BYTES == 128 && sizeof(ncclUniqueId) == 128,
              "include/hjbdp.h hands the communicator id across the C ABI as 128 bytes (hjb_rank_comm_unique_id / _comm_init)");

extern "C" {

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
    ncclComm_t comm = nullptr;
    hipStream_t xfer = nullptr;
    hipEvent_t xready = nullptr;
    double *d_partials = nullptr, *d_sums = nullptr;
    bool loopback = false;
    int dtype = HJB_F32, up_needs = 0, dn_needs = 0;
    int64_t xfer_delay_ticks = 0;     // option "xfer_delay_us": a spin of that length behind every exchange (link-latency emulation)
    bool monitor_single = false;      // option "monitor_single": hjb_rank_sweep's monitor in single precision (see there)
    hipEvent_t xdone = nullptr;       // recorded on the transfer stream behind every exchange (hjb_rank_step_post's strips wait for it)
    bool post_exchange = true;        // option "post_exchange": hjb_rank_sweep runs hjb_rank_step_post (1, default) or hjb_rank_step (0)
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
    if (!strcmp(key, "post_exchange")) { r->post_exchange = value != 0; return HJB_OK; }      // hjb_rank_sweep: hjb_rank_step_post (1) / hjb_rank_step (0)
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

// a separable terminal cost built on the device in this rank's haloed buffer (owned planes AND halo planes: a grid like C3's
// never exists in host memory) - hjb_device_fill_separable on the rank's slab handle
int32_t hjb_rank_fill_separable(hjb_rank r, const void *const *vecs, void *dJ, void *stream) {
    if (!r || !r->whole) return rfail(r, HJB_E_INVALID, "null argument");
    const int st = hjb_device_fill_separable((hjb_handle)r->whole, vecs, dJ, stream);
    if (st) return rfail(r, st, "%s", r->whole->err.c_str());
    return HJB_OK;
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

}  // extern "C"
// strips_first: the boundary strips are enqueued before the interior (their halos are there already: hjb_rank_step_post)
static int rank_stage(hjb_rank r, const void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream, void *halo_stream, bool strips_first) {
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
    int st = HJB_OK;
    if (!strips_first) {
        st = stage_part(0, cs);
        if (st) return st;
    }
    if (halos) RANK_TRY(hipEventRecord(r->halo, hs));
    for (int k = 1; k <= 2; ++k)
        if (r->part[k]) {
            if (halos) RANK_TRY(hipStreamWaitEvent(r->ss[k - 1], r->halo, 0));
            st = stage_part(k, r->ss[k - 1]);
            if (st) return st;
            RANK_TRY(hipEventRecord(r->sdone[k - 1], r->ss[k - 1]));
        }
    if (strips_first) {
        st = stage_part(0, cs);
        if (st) return st;
    }
    for (int k = 1; k <= 2; ++k)
        if (r->part[k]) RANK_TRY(hipStreamWaitEvent(cs, r->sdone[k - 1], 0));
    return HJB_OK;
#undef RANK_TRY
}
extern "C" {

int32_t hjb_rank_stage(hjb_rank r, const void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream, void *halo_stream) {
    return rank_stage(r, dJ_in, dJ_out, d_idx, compute_stream, halo_stream, false);
}

// The pieces of hjb_rank_step_post for a host that moves the halo planes itself (hjbdp/sharded.py with torch.distributed):
// hjb_rank_stage_post = the stage with the boundary strips enqueued FIRST, behind what halo_stream holds (the previous
// exchange); hjb_rank_wait_strips makes `stream` wait for that stage's strips - the caller's exchange of dJ_out's boundary
// planes goes there, under the rest of the interior.  out: *covered = 1 when the strips cover every plane a neighbour needs
// (else the caller's exchange must wait for the whole stage: the compute stream).
int32_t hjb_rank_stage_post(hjb_rank r, const void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream, void *halo_stream) {
    return rank_stage(r, dJ_in, dJ_out, d_idx, compute_stream, halo_stream, true);
}

int32_t hjb_rank_wait_strips(hjb_rank r, void *stream, int32_t *covered) {
    if (!r) return rfail(r, HJB_E_INVALID, "null argument");
    const int owned = r->end - r->begin;
    const int lo_w = r->part[1] ? r->need_lo : 0, hi_w = r->part[2] ? r->need_hi : 0;
    const bool cover = r->part[0] && r->dn_needs <= (r->part[1] ? lo_w : (r->dn_needs ? 0 : owned)) &&
                       r->up_needs <= (r->part[2] ? hi_w : (r->up_needs ? 0 : owned));
    if (covered) *covered = cover ? 1 : 0;
    if (!cover) return HJB_OK;
    if (hipSetDevice(r->device) != hipSuccess) return rfail(r, HJB_E_DEVICE, "hipSetDevice failed");
    for (int k = 0; k < 2; ++k)
        if (r->part[k + 1] && hipStreamWaitEvent((hipStream_t)stream, r->sdone[k], 0) != hipSuccess) return rfail(r, HJB_E_DEVICE, "hipStreamWaitEvent failed");
    if (hipStreamWaitEvent((hipStream_t)stream, r->fork, 0) != hipSuccess) return rfail(r, HJB_E_DEVICE, "hipStreamWaitEvent failed");
    return HJB_OK;
}

// ---- RCCL inside the library: the halo exchange and the monitor's all-reduce of a rank, no torch, no MPI ------------------
// per stage `ncclGroupStart; ncclSend / ncclRecv x <= 4; ncclGroupEnd` on a transfer stream (one xGMI link per
// neighbour pair), every monitor period a 2-double ncclAllReduce.  librccl is dlopen'ed on first use: libhjbdp carries no link
// dependency on it (a single-GPU host never loads it; a process that already holds a librccl - torch's - shares it).
namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;          // optional: what the communicator itself reports (hjb_rank_comm_info)
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    std::string why;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
constexpr ncclDataType_t kNcclUint8 = ncclUint8, kNcclFloat64 = ncclFloat64;
constexpr ncclRedOp_t kNcclSum = ncclSum;

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
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    a.CommCount = (decltype(a.CommCount))dlsym(lib, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(lib, "ncclCommUserRank");
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
        const ncclResult_t e_ = (expr);                                                                       \
        if (e_ != ncclSuccess) return rfail(r, HJB_E_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(e_)); \
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
    if (r->xdone) (void)hipEventDestroy(r->xdone);
    r->xdone = nullptr;
    if (r->d_partials) (void)hipFree(r->d_partials);
    if (r->d_sums) (void)hipFree(r->d_sums);
    r->xfer = nullptr; r->xready = nullptr; r->d_partials = nullptr; r->d_sums = nullptr;
}

int32_t hjb_rank_comm_available(void) {
    if (!rccl_load()) return rfail(nullptr, HJB_E_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why.c_str());
    return HJB_OK;
}

int32_t hjb_rank_comm_unique_id(void *id128_out) {
    if (!id128_out) return rfail(nullptr, HJB_E_INVALID, "null argument");
    if (!rccl_load()) return rfail(nullptr, HJB_E_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why.c_str());
    RCCL_TRY(nullptr, g_rccl.GetUniqueId((ncclUniqueId *)id128_out));
    return HJB_OK;
}

int32_t hjb_rank_comm_init(hjb_rank r, const void *id128) {
    if (!r || !id128) return rfail(r, HJB_E_INVALID, "null argument");
    if (r->comm) return rfail(r, HJB_E_INVALID, "this rank already has a communicator");
    if (!rccl_load()) return rfail(r, HJB_E_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why.c_str());
    RANKH_TRY(r, hipSetDevice(r->device));
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof id.internal);
    // loopback (option "comm_loopback", the one-GPU transport test): a communicator of ONE rank, both neighbours = this rank
    RCCL_TRY(r, g_rccl.CommInitRank(&r->comm, r->loopback ? 1 : r->world, id, r->loopback ? 0 : r->rank));
    hipError_t e = hipStreamCreateWithFlags(&r->xfer, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->xready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->xdone, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(r->xdone, r->xfer);          // "no exchange pending"
    if (e == hipSuccess) e = hipMalloc((void **)&r->d_partials, sizeof(double) * 2 * kReduceBlocks);
    if (e == hipSuccess) e = hipMalloc((void **)&r->d_sums, sizeof(double) * 2);
    if (e != hipSuccess) {          // leave nothing half-built: a retry must not be refused with "already has a communicator"
        rank_comm_release(r);
        return rfail(r, e == hipErrorOutOfMemory ? HJB_E_NOMEM : HJB_E_DEVICE, "hjb_rank_comm_init: %s", hipGetErrorString(e));
    }
    return HJB_OK;
}

// What the communicator ITSELF reports (ncclCommCount / ncclCommUserRank): a multi-GPU run verifies with it that the ranks it
// timed were ranks of one communicator of the expected size (bench.py prints it).  -1 where librccl lacks the query.
int32_t hjb_rank_comm_info(hjb_rank r, int32_t *n_ranks, int32_t *comm_rank) {
    if (!r) return rfail(r, HJB_E_INVALID, "null argument");
    if (!r->comm) return rfail(r, HJB_E_INVALID, "hjb_rank_comm_init first");
    int n = -1, me = -1;
    if (g_rccl.CommCount) RCCL_TRY(r, g_rccl.CommCount(r->comm, &n));
    if (g_rccl.CommUserRank) RCCL_TRY(r, g_rccl.CommUserRank(r->comm, &me));
    if (n_ranks) *n_ranks = n;
    if (comm_rank) *comm_rank = me;
    return HJB_OK;
}

// The halo exchange of dJ (this rank's haloed J buffer) on the library's transfer stream, ordered behind everything
// `compute_stream` holds at the time of the call (the stage that wrote dJ).  Returns at once; hjb_rank_stage's halo_stream
// argument = hjb_rank_transfer_stream(r) makes the boundary strips wait for it (hjb_rank_step does both).
}  // extern "C"
// the grouped send / recv of dJ's boundary planes on the transfer stream, which the caller has ordered behind the stage
// (or the strips) that wrote them; xdone is recorded behind it
static int rank_exchange_on_xfer(hjb_rank r, void *dJ) {
    const size_t plane_b = (size_t)r->inner * r->esz;
    const int owned = r->end - r->begin;
    char *J = (char *)dJ;
    const int dn = r->loopback ? 0 : r->rank - 1, up = r->loopback ? 0 : r->rank + 1;
    if (!(r->dn_needs || r->hlo || r->up_needs || r->hhi)) { RANKH_TRY(r, hipEventRecord(r->xdone, r->xfer)); return HJB_OK; }
    RCCL_TRY(r, g_rccl.GroupStart());
    ncclResult_t e1 = ncclSuccess;
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
    const ncclResult_t e2 = g_rccl.GroupEnd();
    if (e1 != ncclSuccess || e2 != ncclSuccess) return rfail(r, HJB_E_DEVICE, "halo exchange: %s", g_rccl.GetErrorString(e1 != ncclSuccess ? e1 : e2));
    if (r->xfer_delay_ticks > 0) hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, r->xfer, (long long)r->xfer_delay_ticks);
    RANKH_TRY(r, hipEventRecord(r->xdone, r->xfer));
    return HJB_OK;
}
extern "C" {

int32_t hjb_rank_exchange(hjb_rank r, void *dJ, void *compute_stream) {
    if (!r || !dJ) return rfail(r, HJB_E_INVALID, "null argument");
    if (!r->comm) return rfail(r, HJB_E_INVALID, "hjb_rank_comm_init first");
    RANKH_TRY(r, hipSetDevice(r->device));
    RANKH_TRY(r, hipEventRecord(r->xready, (hipStream_t)compute_stream));
    RANKH_TRY(r, hipStreamWaitEvent(r->xfer, r->xready, 0));
    return rank_exchange_on_xfer(r, dJ);
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

// The same stage in the order that hides the exchange completely: a stage's boundary strips need the halos of ITS input,
// and what the neighbours need next is this stage's strips' OUTPUT.  So: the strips first (their halos arrived during the
// previous stage), the interior beside them, and the exchange of dJ_out's boundary planes on the transfer stream as soon as
// the strips are done - it has the whole rest of the interior to complete, and the next stage starts with its halos in
// place.  (hjb_rank_step exchanges dJ_in FIRST: its strips start an exchange late and the next stage's exchange cannot start
// before they end.)  Precondition: dJ_in's halo planes are valid - ONE hjb_rank_exchange(r, dJ_in, compute_stream) before the
// first step; every step leaves dJ_out's halos filled (the transfer may still be in flight when the call returns: the next
// hjb_rank_step_post, hjb_rank_transfer_stream(r) or a device synchronisation orders behind it).  A rank whose neighbours need
// planes its strips do not cover (need_lo != need_hi) sends after its interior; a rank without the split, after its one kernel.
int32_t hjb_rank_step_post(hjb_rank r, void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream) {
    if (!r || !dJ_in || !dJ_out) return rfail(r, HJB_E_INVALID, "null argument");
    const bool comm = r->world > 1 || r->loopback;
    if (!comm) return rank_stage(r, dJ_in, dJ_out, d_idx, compute_stream, nullptr, false);
    if (!r->comm) return rfail(r, HJB_E_INVALID, "hjb_rank_comm_init first");
    RANKH_TRY(r, hipSetDevice(r->device));
    hipStream_t cs = (hipStream_t)compute_stream;
    // the strips (or, without the split, the one kernel) wait for the previous exchange: halo_stream = the transfer stream
    int st = rank_stage(r, dJ_in, dJ_out, d_idx, compute_stream, (void *)r->xfer, true);
    if (st) return st;
    const int owned = r->end - r->begin;
    const int lo_w = r->part[1] ? r->need_lo : 0, hi_w = r->part[2] ? r->need_hi : 0;       // planes the strips cover
    const bool strips_cover = r->part[0] && r->dn_needs <= (r->part[1] ? lo_w : (r->dn_needs ? 0 : owned)) &&
                              r->up_needs <= (r->part[2] ? hi_w : (r->up_needs ? 0 : owned));
    if (strips_cover) {
        for (int k = 0; k < 2; ++k)
            if (r->part[k + 1]) RANKH_TRY(r, hipStreamWaitEvent(r->xfer, r->sdone[k], 0));
        // ... and, like the strips themselves, behind everything the compute stream held before this stage (J_out's halo
        // planes were the previous stage's input)
        RANKH_TRY(r, hipStreamWaitEvent(r->xfer, r->fork, 0));
    } else {
        RANKH_TRY(r, hipEventRecord(r->xready, cs));
        RANKH_TRY(r, hipStreamWaitEvent(r->xfer, r->xready, 0));
    }
    return rank_exchange_on_xfer(r, dJ_out);
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
    // one communicator, one order: the all-reduce goes behind whatever exchange is still pending on the transfer stream
    // (post-exchange order; include/hjbdp.h "concurrent use of the communicator") - once per monitor period
    if (r->xdone) RANKH_TRY(r, hipStreamWaitEvent(cs, r->xdone, 0));
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
    const bool post = r->post_exchange && (r->world > 1 || r->loopback);
    if (post && !st && n_stages >= 1) st = hjb_rank_exchange(r, J[0], compute_stream);      // the terminal cost's halos, once
    for (int k_s = n_stages; k_s >= 1 && !st; --k_s) {
        st = post ? hjb_rank_step_post(r, J[cur], J[1 - cur], d_idx, compute_stream) : hjb_rank_step(r, J[cur], J[1 - cur], d_idx, compute_stream);
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
        // post-exchange order: the last stage's send / recv into the final buffer's halo planes may still be running on the
        // transfer stream (also after an early stop).  The sweep ends - and is timed - behind it: a caller that refills the
        // buffer or hands it back to a stream-ordered allocator must not race with that receive
        if (post && r->xdone && done > 0) (void)hipStreamWaitEvent(cs, r->xdone, 0);
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

}  // extern "C"
