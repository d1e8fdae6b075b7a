// hjbdp_host.h - what the host-side translation units of libhjbdp share (internal; include/hjbdp.h is the public ABI):
// the handle, error reporting, the process-wide locks and switches, and the entry points one unit offers the others.
//   hjbdp_setup.hip    problem upload, stage-invariant tables, kernel choice and plans, the stage launch (hjb_create's work)
//   hjbdp_api.hip      hjb_create .. hjb_solve, options, probe, policy lookup (the single-device C ABI)
//   hjbdp_builder.hip  the flat builder API (MATLAB loadlibrary / calllib)
//   hjbdp_multi.hip    hjb_create_multi / hjb_solve_multi (one process, several GPUs)
//   hjbdp_rank.hip     hjb_rank_* (one process per GPU) and the RCCL transport inside the library
//   hjbdp_devmem.hip   device-buffer helpers
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/hjbdp.h"
#include "hjbdp_dev.h"
#include "hjbdp_launch.h"
#include "kernels_generic.h"
#include "kernels_nested.h"
#include "kernels_packed.h"
#include "kernels_packed2.h"
#include "kernels_uniwin.h"
#include "kernels_ctrlsplit.h"
#include "kernels_tabled.h"
#include "kernels_rowwise.h"
#include "kernels_tile2d.h"
#include "kernels_colsweep.h"
#include "kernels_colcoop.h"
#include "kernels_reduce.h"
#include "kernels_probe.h"

namespace hjbhost {
using namespace hjb;

constexpr int kGraphStages = 32;   // even: a replay starts and ends in dJ[0]

extern thread_local std::string g_last_error;
// fault injection for the tests, by explicit call only (hjb_test_hook; the environment never changes what the library does)
extern std::atomic<int> g_test_fail_tab64_scratch;   // "fail_tab64_scratch": the float64 table build's scratch allocation fails
extern std::atomic<int> g_test_fail_tabled_alloc;    // "fail_tabled_alloc": the (cell, t) table allocation fails
extern std::atomic<int> g_test_rccl_only_env;        // "rccl_only_env": the RCCL loader tries $HJBDP_RCCL_LIB only
// Handles may be driven from different host threads (one thread per handle).  HIP stream capture is fragile
// against "unsafe" calls made elsewhere in the process while it records (device-wide synchronisation, synchronous
// copies, allocation): a capture takes this lock exclusively, every such call takes it shared.  Kernel launches,
// graph launches and waits on a handle's own stream need no lock and overlap freely.
extern std::shared_mutex g_capture_mu;

struct Handle {
    hjb_problem prob{};  // scalar fields only (pointers are not kept)
    int device = 0;
    int dtype = HJB_F32;
    size_t esz = 4;
    int64_t n_owned = 0, nU = 0, j_elems = 0, inner = 0;
    int nplanes = 0, plane0 = 0;
    DParams hp{};                 // host copy of the device params
    DParams *dp = nullptr;        // device params
    std::vector<void *> allocs;   // every device allocation (freed in destroy): large ones and the chunks the small ones are carved from
    char *arena = nullptr;        // dev_alloc: the current chunk's free part
    size_t arena_left = 0;
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
    // K15 (kernels_uniwin.h, variant 4 modes 7 / 8): the window kernel for chunks that share their rate axes
    bool uniwin_ok = false;       // the structure holds and the per-point plan is built
    bool uniwin_auto = false;     // ... and few enough points leave the usual shape for it to be the automatic choice
    int uniwin_on = -1;           // option "uniwin": -1 automatic, 0 never, 1 whenever uniwin_ok
    int uniwin_slow = 0;          // points of the plan that take the slow path
    int uw_tile = 0;              // option "uw_tile": log2 tile extents lA + 8 * lB + 64 * lC (0: the default 3, 2, 2)
    int uw_grid = 0;
    int uw_claim = 1;             // option "uw_claim": 1 = the chunk walk's positions are claimed from per-XCD counters, 0 = fixed stride
    int uw_block = 256;           // option "uw_block": states per chunk = threads per workgroup (256 or 64)
    size_t uw_lds = 0;
    DUniwin huw{};
    DUniwin *duw = nullptr;
    bool tabled_ok = false;       // variant 5: per-axis (cell, t) tables for every axis (built on first use)
    bool tabled_i32 = false;      // ... and every index of it fits 31 bits: the 32-bit form of the kernel runs (kernels_tabled.h)
    bool tabled_i32_on = true;    // option "tabled_i32" (0: the 64-bit form anyway - A/B timing, tests)
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

int fail(Handle *h, int code, const char *fmt, ...);

// Work of hjb_create / hjb_solve's set-up (table builds, plans, memsets) is issued on the NULL stream; a handle's sweep runs on its own
// non-blocking stream.  Waiting for the set-up is a wait on the null stream - NOT hipDeviceSynchronize, which also waits for every other
// handle's sweep in flight (independent channels solved side by side from several host threads, hjbdp.core.solve_many, ran one after
// the other for it).
inline hipError_t sync_setup() { return hipStreamSynchronize(nullptr); }

#define HIP_TRY(h, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, HJB_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

int dev_alloc(Handle *h, size_t bytes, void **out);
template <typename T>
int upload(Handle *h, const std::vector<T> &v, void **out) {
    void *d = nullptr;
    const int ast = dev_alloc(h, std::max<size_t>(v.size(), 1) * sizeof(T), &d);
    if (ast) return ast;
    if (!v.empty()) HIP_TRY(h, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = d;
    return HJB_OK;
}

int dev_alloc(Handle *h, size_t bytes, void **out);

// hjbdp_setup.hip
int64_t term_elems(const hjb_problem *p, uint32_t mask);
int build_handle(Handle *h, const hjb_problem *p);       // upload + analysis of a validated problem (float32 / float64 arithmetic by p->dtype)
void halo_of_problem(const hjb_problem *p, bool tab64, int *lo, int *hi);     // the halo the last axis' terms imply
int colsweep_map(Handle *h, const std::vector<int32_t> &plan);
int colsweep_dpp_ok_f32(Handle *h, bool *ok);
int ensure_axis0_table(Handle *h);
int ensure_tabled(Handle *h);
int rebuild_tables(Handle *h, bool mfma);
int table_hash(Handle *h, uint64_t *out);
int ensure_colsweep(Handle *h);
void uniwin_tiles(Handle *h);        // workgroup size, chunk count, tile extents, LDS and launch grid -> Handle (the caller uploads Handle::huw)
inline bool uniwin_active(const Handle *h) {
    return h->uniwin_ok && (h->packed_pre == 5 || h->packed_pre == 6) && (h->uniwin_on == 1 || (h->uniwin_on < 0 && h->uniwin_auto));
}
void colsweep_split(Handle *h);
int colsweep_upload(Handle *h);      // the device copy of Handle::hcs, launch record included
int examine_tile2d(Handle *h);
int launch_tile2d(Handle *h, const void *dJn, void *dJo, void *didx, int K, hipStream_t st);
void choose_launch(Handle *h);
int launch_stage(Handle *h, const void *dJn, void *dJo, void *didx, hipStream_t st);
int ensure_work(Handle *h);
int check_status(Handle *h, hipStream_t st);
int make_probe(Handle *h, const hjb_probe *pb, DProbe *out);
int launch_probe(Handle *h, const DProbe &pr, const void *dJn, hipStream_t st);

// First element of a term / table array that is not finite (-1: all finite).  The kernels' contract covers finite data only
// (DESIGN.md section 2): a NaN in a table is refused where it enters, with its place named, not found in J 2000 stages later.
int64_t first_nonfinite(const void *data, int64_t n, bool f64);
constexpr int64_t kMaxStates = (int64_t)1 << 40;     // more grid points than any device of this generation can hold one byte for

// hjbdp_api.hip: everything hjb_create checks or derives WITHOUT touching a device (the partitioners use it)
int analyse_problem(const hjb_problem *p, int *idx_bytes_out, int64_t *n_states_out, int *halo_lo, int *halo_hi);

}  // namespace hjbhost

// hjbdp_builder.hip: the flat builder (hjb_problem_new ...) and its problem with the pointers bound
struct hjb_builder_s {
    hjb_problem p{};
    std::vector<std::vector<double>> knots;
    std::vector<std::vector<unsigned char>> blobs;   // owned copies of every term / model table
    std::string err;
};
extern "C" {
int builder_bind(hjb_builder b, hjb_problem *out);
int bfail(hjb_builder b, int code, const char *fmt, ...);
}
