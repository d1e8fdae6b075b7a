// kernels_colsweep.h - variant 7: the column-sweep stage kernel for the pos-att shape (C4 / C5).
//
// Shape (checked on the host, hjbdp_setup.hip::ensure_colsweep): D = 4, one control dim, and
//   * axes 0 and 1 do not depend on the control (pos-att with the axes relabelled (x, theta, v, w):
//     x+ = x + h v over (x, v), theta+ = theta + h w over (theta, w) - Solver_pos_att.m:299-328);
//     axis 0's cell does not depend on state dim 1, axis 1's cell does not depend on state dim 0;
//   * axes 2 and 3 depend on state dims 2, 3 and the control only (v+ over (v, u), w+ over (w, u)).
// The canonical lerp order (axis 0 first ... axis 3 last) then lets the two control-INDEPENDENT lerps
// be done once per (state, corner row) instead of once per (state, control, corner row):
//
//   A[k1; k2, k3]  = lerp_axis0( J[c0, k1, k2, k3], J[c0+1, k1, k2, k3]; t0 )      c0, t0: the thread's own (i0, i2, i3)
//   B[k2, k3]      = lerp_axis1( A[c1; ..], A[c1+1; ..]; t1 )                       c1, t1: uniform over the wave
//   total_u        = g_u + lerp_axis3( lerp_axis2( B[c2_u + {0,1}, c3_u + {0,1}] ) )
//
// and A does not depend on the state's index along axis 1.  So one WAVE = consecutive axis-0 states of one (i2, i3)
// pair, and it SWEEPS THE COLUMN i1 = 0..n1-1: the A row at knot c1+1 of one step is the A row at knot c1 of the next
// (axis-1 cells advance by one per state when the displacement is sub-cell; anything else re-primes, a wave-uniform
// branch).  Per step a lane loads each needed corner row once, not once per control.
//
// Which (k2, k3) rows a state needs is stage-invariant and identical for the whole column, so the host builds a PLAN
// per (i2, i3): the controls are put into GROUPS sharing the cell of the "group axis" GAX (pos-att: w, 5 distinct
// cells among the 9 thruster combinations, Solver_pos_att.m:886-904) and a window of 3 knots of the other, "window"
// axis (v moves < 1 cell).  A group = 2 x 3 corner rows and 6 member SLOTS: slots 0-2 interpolate between window
// knots (0, 1), slots 3-5 between (1, 2) - which pair a control uses is therefore static, and a slot's weights,
// control cost and control number are wave-uniform plan data parked in LDS for the column.  The kernel is a template
// over the number of groups NG (the maximum over the plans; plans with fewer are padded with member-less groups) and
// straight-line over groups, rows and slots: the rolling A values sit in registers with static indices.  Groups are
// visited in plan order, not control order.
// Slots of a pair are filled from its first one, so a pair's first slot test is the pair's test and an empty slot ends
// it; a window knot no slot uses (26 of 30 rows remain on C4) is neither gathered nor lerped; a slot visited after a
// higher-numbered control takes over a TIE first (first index wins, exactly as MATLAB's min).
//
// The two axis-0 neighbours (c0, c0 + 1) of a corner row: the first version loaded both (profiles/
// r02_c4_colsweep_v1_pmc.json: TA busy 87 %, 2.74 ms per stage), and the two loads of a pair hit the same cache lines.
// One-load form (template DPP, chosen on the host when every wave's axis-0 cells satisfy it): a wave carries 60 states,
// lane L loads knot kb + L ONCE, and a state's upper neighbour is lane L + 1's value, taken as the DPP operand of the
// subtraction (`v_sub_f32_dpp ... wave_shl:1`; lane 60 is the halo loader).  A wave with ONE state whose cell sits
// elsewhere (grid-edge clamping: half of C4's waves) gives that state the spare lane pair 61, 62, so every state lane
// is regular; the host admits the form when no 60-state chunk has more than one such state.
// (Fetching the neighbour through the LDS crossbar, ds_bpermute_b32, was measured slower: 2.90 vs 2.67 ms at the time.)
//
// The rest of the schedule, each step of it measured (profiles/r0*_c4_experiments.log): gathers issued
// and awaited by hand in two halves of the group sequence (counted s_waitcnt through a computed jump); results stored
// at once by hand-issued stores (HJB_CS_DIRECT below; parked in LDS and flushed every 16 steps until round 5); the four
// waves of a workgroup - neighbours on the group axis - take every step together (s_barrier) so that they share L1
// lines; a launch with few columns (a multi-GPU slab, its boundary strips, a small grid) sweeps each column in several
// parts, one wave each (DColSweep::split); workgroup b serves XCD b % 8.  C4 (120^4 x 9, float32): 1.52 - 1.60 ms per
// stage by box = 1.17 - 1.23e12 backups/s (round 2: 1.81).  Same values, same arithmetic as every other variant:
// bit-identical results.
#pragma once
#include <cstddef>
#include "hjbdp_dev.h"
#include "kernels_generic.h"
#include "kernels_tabled.h"
#include "kernels_rowwise.h"

namespace hjb {

constexpr int kCsGMax = 6;      // groups per (i2, i3)
constexpr int kCsMMax = 6;      // member slots per group: 3 per window pair
constexpr int kCsNW = 3;        // window knots per group
constexpr int kCsUMax = 16;     // controls
constexpr int kCsMaxCu = kLeanMaxCu;
// Results are parked in LDS and written out every kCsFlush steps.  16 steps (25 KB of LDS per workgroup, was 20 steps / 30 KB) leave room
// for SIX workgroups per CU, and the usual cost shape (FASTCOST, float32 J storage; with five groups: group axis 2 only - axis 3 would spill seven
// registers, the binary16 form fifteen)
// fits 80 registers: six waves per SIMD instead of five, C4 1.686 -> 1.655 ms
// on one box (profiles/r04_c4_experiments.log).  A/B: -DHJB_CS_FLUSH=20 -DHJB_CS_WAVES=5.
#ifndef HJB_CS_FLUSH
#define HJB_CS_FLUSH 16
#endif
#ifndef HJB_CS_WAVES
#define HJB_CS_WAVES 6
#endif
constexpr int kCsFlush = HJB_CS_FLUSH;
// HJB_CS_DIRECT (round 5): every step STORES its result at once instead of parking 16 steps in LDS and writing them out behind
// a full drain.  The drain was there because gfx9 counts loads and stores on one counter and they complete out of order WITH
// EACH OTHER; but the counted waits stay safe with stores pending as long as the count they wait for is the number of younger
// LOADS only: vmcnt <= N leaves at most N - (stores pending) loads pending, and loads complete in order among themselves, so the
// awaited (older) gathers have landed whatever the stores do - a store can only make a wait last longer, never let it pass early.
#ifndef HJB_CS_DIRECT
#define HJB_CS_DIRECT 1
#endif
constexpr int kCsDppLanes = 60; // states per wave in the one-load form (+ a halo lane + a spare lane pair + 1)
// The plan of one (i2, i3), 32-bit words:
//   [0] halo violation flag | groups << 8   [1 + g] byte offset of group g's first corner row
//   [1 + GMAX + g] bit s = slot s is used, bit 16 + s = slot s is visited after a higher-numbered control
//   [kCsPI + 8 s ...] member slot s' = g * MMAX + s: t_window, t_group, cu[0], control number, cu[1..3], 0
//   [1 + 2 GMAX + g] cooperative form (kernels_colcoop.h): byte offset of group g's first corner row in the staged rows
constexpr int kCsPI = 24;
constexpr int kCsSlots = kCsGMax * kCsMMax;
constexpr int kCsPlanWords = kCsPI + 8 * kCsSlots;
static_assert(1 + 3 * kCsGMax <= kCsPI && kCsPI % 4 == 0, "plan header; the slots are read as 16-byte vectors");

// Every scalar a wave needs before it knows its column, as ONE record of 48 words at the head of DColSweep: the kernel reads
// it with three s_load_dwordx16 and one wait (k_backup_colsweep, "hop 1").  Filled on the host from DParams / DTabled /
// DColSweep themselves (hjbdp_setup.hip::colsweep_upload) - a copy for speed, never a second source of truth.
enum CsRec {
    kRecXcdCnt = 0,                                   // [8]
    kRecN0 = 8, kRecN1, kRecN2, kRecN3,
    kRecSplit = 12, kRecWin, kRecXStride, kRecNcu,
    kRecXcdIg = 16,                                   // pointer (2 words)
    kRecPlan = 18,                                    // pointer
    kRecA0Tab = 20, kRecA1Tab = 22, kRecStatus = 24,  // pointers
    kRecGBytes = 26, kRecWBytes, kRecS1Bytes,
    kRecNpreCol = 29, kRecNpre, kRecStepUniform,
    kRecA0S0 = 32, kRecA0S2, kRecA0S3, kRecA1S1, kRecA1S2, kRecA1S3,
    kRecSlabBegin = 38, kRecHaloLo, kRecJs1, kRecJs2, kRecJs3, kRecIndexBase, kRecIdxBytes,
    kRecWords = 48
};
// ... and the cost terms a column reads before its first step, for the usual shapes: up to three column-constant state terms
// and the one per-step term, each {data pointer, three strides}: 24 words read beside the plan header ("hop 3").
enum CsCostRec { kCRecNCol = 0, kCRecHasSu = 1, kCRecTerm = 2 /* 5 words each x 3 */, kCRecSu = 17 /* 5 words */, kCRecWords = 24 };
struct DColSweep {
    alignas(64) uint32_t rec[kRecWords];
    uint32_t crec[kCRecWords];
    const int32_t *plan;
    int32_t gax;            // group axis: 2 or 3 (the other one is the window axis)
    int32_t ng;             // groups per plan (maximum over the plans)
    int32_t npre_col;       // leading state-only cost terms that do not depend on state dim 1: summed once per column
    int32_t step_uniform;   // the remaining state-only cost terms do not depend on state dim 0 (wave-uniform per step)
    int32_t ncu;            // control-only cost terms
    int32_t dpp;            // one-load form: every wave of kCsDppLanes states shares one (cell - index) but for one state at most
    uint32_t g_bytes;       // J byte stride of the group axis
    uint32_t w_bytes;       // J byte stride of the window axis
    uint32_t s1_bytes;      // J byte stride of axis 1
    // Which column a wave takes (XCD-aware: workgroup b runs on XCD b % 8, each XCD has its own L2).  A corner row is
    // shared by the columns whose group-axis index differs by a multiple of the spacing of the groups (pos-att: w moves
    // 3.8 cells per thruster level, so columns i3, i3 + 4, i3 + 8 ... gather from the same planes) and by neighbours along
    // the window axis.  Each XCD therefore owns the group-axis indices of (part of) one residue class, xcd_ig[x][0..cnt),
    // and walks them fastest, then the axis-0 chunk, then the window-axis index.
    const int32_t *xcd_ig;  // [8][xcd_stride]
    int32_t xcd_cnt[8];
    int32_t xcd_stride;
    int32_t xcd_win;        // the XCDs split the WINDOW axis instead (xcd_ig then lists window-axis indices)
    int32_t split;          // a column is swept by `split` waves, each a contiguous part of i1 (priming where it starts): few
                            // columns (a boundary strip of a multi-GPU slab, a small grid) still fill the chip, and a launch
                            // lasts n1 / split steps instead of n1
    int32_t coop;           // cooperative form (kernels_colcoop.h): 0 = off, else elements per staging load (1, 4, 8)
    const int32_t *wg;      // its per-workgroup words, [(ig * chunks + chunk) * blocks + block][kCcWgWords]
};

__device__ __forceinline__ float lane_up(float x) {      // value of lane + 1 (lane 63 reads 0: a halo lane, never used)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}
// (value of lane + 1) - (own value): written so that the DPP move folds into the subtraction (v_sub_f32_dpp)
__device__ __forceinline__ float lane_up_minus(float x) {
    float d = lane_up(x) - x;
    asm volatile("" : "+v"(d));          // stays a scalar subtraction: packed with its twin it could not take the DPP operand
    return d;
}
// The running minimum over the controls, (best, best_u) <- (tot, u) if tot < best; a slot visited after a higher-numbered
// control first takes over a TIE (take_tie).  Written out because hipcc turns the wave-uniform choice between the two
// forms into selects and computes every compare for every slot.
__device__ __forceinline__ void take_less(float &best, int &best_u, float tot, int u) {          // tot < best
    asm volatile("v_cmp_lt_f32 vcc, %2, %0\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc"
                 : "+v"(best), "+v"(best_u) : "v"(tot), "v"(u) : "vcc");
}
__device__ __forceinline__ void take_tie(float best, int &best_u, float tot, int u) {      // tot == best and u < best_u
    unsigned long long m;
    asm volatile("v_cmp_eq_f32 %1, %2, %4\n\tv_cmp_lt_i32 vcc, %3, %0\n\ts_and_b64 vcc, vcc, %1\n\t"
                 "v_cndmask_b32 %0, %0, %3, vcc"
                 : "+v"(best_u), "=&s"(m) : "v"(tot), "v"(u), "v"(best) : "vcc", "scc");     // s_and_b64 writes SCC: without
                 // the clobber a flush test (s_cmp ... s_cselect) scheduled around this block reads a stale SCC - found in round 3
                 // when a restructured build moved one there (wrong write-out timing, garbage labels)
}
// The corner-row gathers of the column loop are issued by hand and waited for by hand.  hipcc's wait insertion is
// path-insensitive: a load that sits behind a wave-uniform guard (a column has ng <= NG groups), or any store that may
// still be pending (gfx9 has ONE counter for loads and stores, and they complete out of order with each other), turns
// every wait into `vmcnt(0)` - a full drain that also waits for the rows just requested for the next half step.  Here the
// loads are asm statements the compiler does not track; a wait is `s_waitcnt vmcnt(N)` with N = the exact number of
// younger gathers, tied to the awaited registers by "+v" operands so that no use can move above it.  Safe because
// loads return in order among themselves, whatever the stores of the loop do (HJB_CS_DIRECT above).
template <int BYTES>
__device__ __forceinline__ uint32_t gather_async(uint32_t off, gptr<char> base) {
    uint32_t r;
    if (BYTES == 4) asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(off), "s"(base));
    else asm volatile("global_load_ushort %0, %1, %2" : "=v"(r) : "v"(off), "s"(base));
    return r;
}
// A result stored the same way: one 32-bit per-lane byte offset against a scalar base, issued by hand (HJB_CS_DIRECT).
template <int BYTES>
__device__ __forceinline__ void store_async(uint32_t off, uint32_t v, __attribute__((address_space(1))) char *base) {
    if (BYTES == 4) asm volatile("global_store_dword %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
    else if (BYTES == 2) asm volatile("global_store_short %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
    else asm volatile("global_store_byte %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_gathers() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N)); }
// The same with a count known only at run time (a multiple of K, at most 9 K): `s_waitcnt` takes an immediate, and a
// switch over ten of them compiles to a chain of some twenty scalar instructions - twice per step.  Instead: a computed
// jump into a table of (s_waitcnt, s_branch) pairs, 8 bytes each; s_getpc_b64 yields the address of the s_add_u32.
template <int K> __device__ __forceinline__ void wait_gathers_n(int younger) {
    static_assert(K == 2 || K == 4, "gathers per window knot");
    const int off = 12 + 8 * (younger / K);          // past s_add_u32, s_addc_u32, s_setpc_b64; then 8 bytes per entry
    if (K == 2)
        asm volatile("s_getpc_b64 vcc\n\ts_add_u32 vcc_lo, vcc_lo, %0\n\ts_addc_u32 vcc_hi, vcc_hi, 0\n\ts_setpc_b64 vcc\n\t"
                     "s_waitcnt vmcnt(0)\n\ts_branch .Lwg%=\n\ts_waitcnt vmcnt(2)\n\ts_branch .Lwg%=\n\t"
                     "s_waitcnt vmcnt(4)\n\ts_branch .Lwg%=\n\ts_waitcnt vmcnt(6)\n\ts_branch .Lwg%=\n\t"
                     "s_waitcnt vmcnt(8)\n\ts_branch .Lwg%=\n\ts_waitcnt vmcnt(10)\n\ts_branch .Lwg%=\n\t"
                     "s_waitcnt vmcnt(12)\n\ts_branch .Lwg%=\n\ts_waitcnt vmcnt(14)\n\ts_branch .Lwg%=\n\t"
                     "s_waitcnt vmcnt(16)\n\ts_branch .Lwg%=\n\ts_waitcnt vmcnt(18)\n.Lwg%=:"
                     : : "s"(off) : "vcc", "scc", "memory");
    else
        asm volatile("s_getpc_b64 vcc\n\ts_add_u32 vcc_lo, vcc_lo, %0\n\ts_addc_u32 vcc_hi, vcc_hi, 0\n\ts_setpc_b64 vcc\n\t"
                     "s_waitcnt vmcnt(0)\n\ts_branch .Lwg%=\n\ts_waitcnt vmcnt(4)\n\ts_branch .Lwg%=\n\t"
                     "s_waitcnt vmcnt(8)\n\ts_branch .Lwg%=\n\ts_waitcnt vmcnt(12)\n\ts_branch .Lwg%=\n\t"
                     "s_waitcnt vmcnt(16)\n\ts_branch .Lwg%=\n\ts_waitcnt vmcnt(20)\n\ts_branch .Lwg%=\n\t"
                     "s_waitcnt vmcnt(24)\n\ts_branch .Lwg%=\n\ts_waitcnt vmcnt(28)\n\ts_branch .Lwg%=\n\t"
                     "s_waitcnt vmcnt(32)\n\ts_branch .Lwg%=\n\ts_waitcnt vmcnt(36)\n.Lwg%=:"
                     : : "s"(off) : "vcc", "scc", "memory");
}
__device__ __forceinline__ void tie6(uint32_t (&r)[2][kCsNW]) {
    static_assert(kCsNW == 3, "six registers per group and neighbour");
    asm volatile("" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[1][2]));
}
typedef float cs_pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tie3(cs_pair (&r)[kCsNW]) {       // the same for three row pairs (the copy-free roll)
    static_assert(kCsNW == 3, "three window knots per group");
    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]));
}
template <typename T, typename TJ> __device__ __forceinline__ T raw_to(uint32_t r) {
    if (sizeof(TJ) == 4) return __uint_as_float(r);
    return (T)__builtin_bit_cast(_Float16, (unsigned short)r);
}

// One group of one step, shared by the column-sweep kernels: roll the group's three window knots (the axis-1 lerp of the
// new corner rows with the previous ones), then visit its member slots.  row0(w) returns, for window knot w, the
// (lower, upper group row) pair of axis-0 lerps of the new corner rows - each kernel gets at the neighbours its own way.
// Aold = those lerps of the previous step, Anew receives this step's: the same array in both kernels.  (Two arrays
// swapping roles in a loop unrolled by two would save the copy of each row, but keep both sets alive: 92 -> 124 VGPRs.)
typedef float cs_f4 __attribute__((ext_vector_type(4)));
typedef float cs_f2 __attribute__((ext_vector_type(2)));
template <typename T, int GAX, bool FASTCOST, typename FA, typename SP, bool C64 = false>
__device__ __forceinline__ void cs_group(int ug, int g, FA row0, const cs_f2 (&Aold)[kCsNW], cs_f2 (&Anew)[kCsNW], T t1, SP slots,
                                         T gstep, int ncu, int npre, T &best, int &best_u, double gstep64 = 0.0) {
    typedef cs_f2 f2;
    typedef cs_f4 f4;
    constexpr int NW = kCsNW, MM = kCsMMax;
    // the slot bits as a value of THIS block: tested with s_bitcmp; hoisted out of the loop each test would become a
    // 64-bit lane mask held in two scalar registers for the whole column
    asm volatile("" : "+s"(ug));
    const f2 t1p = {t1, t1};
    // knot 1 of the window serves both pairs; knot 0 only pair 0's slots, knot 2 only pair 1's
    f2 Bv[NW];
    auto roll = [&](int w) {
        const f2 an = row0(w);
        Bv[w] = __builtin_elementwise_fma(t1p, an - Aold[w], Aold[w]);
        Anew[w] = an;
    };
    // the member-independent half of a member's first lerp:
    //   GAX == 3 (window = axis 2 is lerped first, both group rows at once): Ew[p] = B[p + 1] - B[p]
    //   GAX == 2 (group = axis 2 is lerped first): Dg[w] = B[w].upper - B[w].lower
    f2 Ew[2];
    T Dg[NW];
    roll(1);
    if (GAX == 2) Dg[1] = (T)(Bv[1].y - Bv[1].x);
    auto member = [&](int s) {
        const int off = s / (MM / 2);                // slots 0-2: window knots (0, 1); slots 3-5: (1, 2)
        const f4 ms = slots[(g * MM + s) * 2];       // broadcast read of the slot's plan data
        const T tw = ms.x, tg = ms.y;
        const int u = __float_as_int(ms.w);
        T interp;
        if (GAX == 3) {
            const f2 twp = {tw, tw};
            const f2 v = __builtin_elementwise_fma(twp, Ew[off], Bv[off]);
            interp = fma_t<T>(tg, (T)(v.y - v.x), v.x);
        } else {
            T v0 = fma_t<T>(tg, Dg[off], Bv[off].x);
            asm volatile("" : "+v"(v0));             // two plain fmas: packed, their operands would need moving
            const T v1 = fma_t<T>(tg, Dg[off + 1], Bv[off + 1].x);
            interp = fma_t<T>(tw, (T)(v1 - v0), v0);
        }
        T gg;
        if (FASTCOST && C64) {                       // cost_dtype F64: state part + the control term in double, one rounding
            const f4 mx = slots[(g * MM + s) * 2 + 1];
            const double cu = __hiloint2double(__float_as_int(mx.y), __float_as_int(mx.x));
            gg = (T)(gstep64 + cu);
        } else if (FASTCOST) {                       // the usual shape: state terms + ONE control term
            gg = (T)(gstep + ms.z);
        } else {
            const f4 mx = slots[(g * MM + s) * 2 + 1];
            gg = gstep;
            for (int k = 0; k < ncu; ++k) {
                const T x = k == 0 ? ms.z : (k == 1 ? mx.x : (k == 2 ? mx.y : mx.z));
                gg = (npre == 0 && k == 0) ? x : (T)(gg + x);
            }
        }
        const T tot = (T)(gg + interp);
        // groups are not visited in control order: a slot that comes after a higher-numbered control (flag from the
        // plan) also wins a tie if its control number is the lower one: first index wins, exactly
        if (ug & (0x10000 << s)) take_tie(best, best_u, tot, u);
        take_less(best, best_u, tot, u);
    };
    // the slots of a pair are filled from its first one (the plan builder's order): an empty first slot means the pair -
    // and its outer window knot - is not in use, an empty slot ends the pair
    static_assert(MM == 6, "three slots per window pair");
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (ug & (1 << (3 * p))) {
            roll(2 * p);
            if (GAX == 3) Ew[p] = Bv[p + 1] - Bv[p];
            else Dg[2 * p] = (T)(Bv[2 * p].y - Bv[2 * p].x);
            member(3 * p);
            if (ug & (2 << (3 * p))) {
                member(3 * p + 1);
                if (ug & (4 << (3 * p))) member(3 * p + 2);
            }
        }
    }
}

// Five waves per SIMD (<= 96 VGPRs) for the one-load form with up to five groups - the C4 / C5 kernel, which the
// register allocator otherwise leaves at 98; the wider forms take what they need.
//
// THE COPY-FREE ROLL (round 5; one-load form with float32 J storage, HJB_CS_ROLL2 = 0 builds the round-4 loop).  The A row at
// knot c1 + 1 of one step is the A row at knot c1 of the next.  Written as `A[g][w] = an` after the axis-1 lerp, each of the 13
// row pairs of a step cost a v_mov_b64 (7 % of the step's vector instructions): `an` cannot be formed in A's register, which the
// lerp still reads.  Round 4's loop unrolled by two with two sets of A registers removed the copies and needed 128 VGPRs; left
// to itself the allocator spreads the rows of a loop with two textual steps over twice the registers (and spills 173 at 80).
// This form needs NO register beyond the round-4 loop's and leaves the allocator no choice: per (group, window knot) a step
// holds two register pairs anyway - the pair the gather delivers and the previous step's A pair - and they SWAP ROLES every
// step.  Step p: the gathered pair V[p] becomes the new A row in place (`an = fma(t0, next lane - own, own)` overwrites its
// own operand), the old row V[1 - p] is read by the axis-1 lerp for the last time, and the NEXT step's gather is issued into
// V[1 - p] - a "+v" operand of the gather's asm statement, so it lands in exactly the registers the old row occupied.  Two
// textual copies of the step (kernels_colsweep_step.inc) with p = 0 / 1 make the names compile-time.  Rows only some columns
// use keep their two pairs throughout, as before.  tests/test_kernel_budget.py holds the result to 80 VGPRs and to "no gather
// destination is touched in flight".
#ifndef HJB_CS_ROLL2
#define HJB_CS_ROLL2 0
#endif
template <typename T, typename TJ, int GAX, int NG, bool FASTCOST, bool DPP, bool C64 = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(
    (DPP && NG <= 5 && !C64) ? ((FASTCOST && sizeof(TJ) == 4 && (GAX == 2 || NG <= 4)) ? HJB_CS_WAVES : 5) : 1,
    (DPP && NG <= 5 && !C64) ? ((FASTCOST && sizeof(TJ) == 4 && (GAX == 2 || NG <= 4)) ? HJB_CS_WAVES : 5) : 4)))
k_backup_colsweep(const DParams *__restrict__ P, const DTabled *__restrict__ TB, const DColSweep *__restrict__ CS,
                  const TJ *__restrict__ Jn, TJ *__restrict__ Jout, void *__restrict__ idx_out) {
#define CS_BX blockIdx.x
#include "kernels_colsweep_body.inc"
#undef CS_BX
}

// ---- several problems, one launch (hjb_solve_batch: Solver_pos_att.simplified_run's four channels, pos-att/Solver_pos_att.m:197-242) --
// blockIdx.y = the problem (DCsBatch: kernels_tabled.h); the body is the stage kernel's own.
template <typename T, typename TJ, int GAX, int NG, bool FASTCOST, bool DPP, bool C64 = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(
    (DPP && NG <= 5 && !C64) ? ((FASTCOST && sizeof(TJ) == 4 && (GAX == 2 || NG <= 4)) ? HJB_CS_WAVES : 5) : 1,
    (DPP && NG <= 5 && !C64) ? ((FASTCOST && sizeof(TJ) == 4 && (GAX == 2 || NG <= 4)) ? HJB_CS_WAVES : 5) : 4)))
k_backup_colsweep_batch(const DCsBatch *__restrict__ B, uint32_t mask, int parity) {
    const unsigned ch = blockIdx.y;
    if (!((mask >> ch) & 1u) || blockIdx.x >= B->grid[ch]) return;
    const DParams *__restrict__ P = B->P[ch];
    const DTabled *__restrict__ TB = B->TB[ch];
    const DColSweep *__restrict__ CS = B->CS[ch];
    const TJ *__restrict__ Jn = (const TJ *)B->J[ch][parity];
    TJ *__restrict__ Jout = (TJ *)B->J[ch][parity ^ 1];
    void *__restrict__ idx_out = B->idx[ch];
#define CS_BX blockIdx.x
#include "kernels_colsweep_body.inc"
#undef CS_BX
}

}  // namespace hjb