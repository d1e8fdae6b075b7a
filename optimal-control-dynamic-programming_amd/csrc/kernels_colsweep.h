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
// The rest of the schedule, each step of it measured (profiles/r02_c4_experiments.log): gathers issued
// and awaited by hand in two halves of the group sequence (counted s_waitcnt through a computed jump); results parked
// in LDS and written out every 20 steps; the four waves of a workgroup - neighbours on the group axis - take every
// step together (s_barrier) so that they share L1 lines; a launch with few columns (a multi-GPU slab, its boundary
// strips, a small grid) sweeps each column in several parts, one wave each (DColSweep::split); workgroup b serves XCD
// b % 8.  C4 (120^4 x 9, float32): 1.81 ms per stage = 1.03e12 backups/s.  Same values, same arithmetic as every
// other variant: bit-identical results.
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
// loads return in order and the only stores of the loop are followed by a full drain (see the flush below).
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
    static_assert(sizeof(T) == 4, "float32 arithmetic");
    constexpr int D = 4, NW = kCsNW, MM = kCsMMax, LANES = DPP ? kCsDppLanes : 64;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    __shared__ f4 s_slots[4][kCsSlots * 2];
    __shared__ T s_best[4][HJB_CS_DIRECT ? 1 : kCsFlush][HJB_CS_DIRECT ? 1 : 64];
    __shared__ uint8_t s_idx[4][HJB_CS_DIRECT ? 1 : kCsFlush][HJB_CS_DIRECT ? 1 : 64];      // control numbers (< kCsUMax) as bytes: more steps per flush
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // ---- hop 1: EVERY scalar of the launch in one batch ------------------------------------------------------------------
    // A wave's set-up is a chain of dependent loads - launch scalars -> which column -> the column's plan, table entries and cost
    // terms -> its first rows - and on a small grid, a multi-GPU strip or a short part of a column that chain IS the stage time.
    // Read where they were used through P / TB / CS, the scalars cost a round trip to the scalar cache EACH (seventeen waits before
    // the first gather in the round-4 assembly; hipcc sinks a scalar load to its use whatever the source order).  Here: the record
    // at the head of DColSweep, three loads and ONE wait, by hand.
    typedef uint32_t u16v __attribute__((ext_vector_type(16)));
    u16v rA, rB, rC;
    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx16 %2, %3, 0x80\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(rA), "=&s"(rB), "=&s"(rC) : "s"(CS) : "memory");      // early clobber: a destination must not take the address registers a later load of the batch reads
#define CS_REC(i) ((i) < 16 ? rA[(i) & 15] : ((i) < 32 ? rB[(i) & 15] : rC[(i) & 15]))
#define CS_REC_PTR(TY, i) ((TY)(uintptr_t)((uint64_t)CS_REC(i) | ((uint64_t)CS_REC((i) + 1) << 32)))
    const int n0 = (int)CS_REC(kRecN0), n1 = (int)CS_REC(kRecN1), n2 = (int)CS_REC(kRecN2), n3 = (int)CS_REC(kRecN3);
    const unsigned xcd = blockIdx.x & 7u;
    unsigned cnt = rA[0];
#pragma unroll
    for (int x = 1; x < 8; ++x) cnt = xcd == (unsigned)x ? rA[x] : cnt;
    const int cs_split = (int)CS_REC(kRecSplit), cs_win = (int)CS_REC(kRecWin), cs_xstride = (int)CS_REC(kRecXStride);
    cptr<int32_t> xcd_tab = as_const<int32_t>(CS_REC_PTR(const int32_t *, kRecXcdIg));
    const int32_t *const plan_base = CS_REC_PTR(const int32_t *, kRecPlan);
    const uint32_t g_bytes = CS_REC(kRecGBytes), w_bytes = CS_REC(kRecWBytes), s1_bytes = CS_REC(kRecS1Bytes);
    const int ncu = (int)CS_REC(kRecNcu), npre_col = (int)CS_REC(kRecNpreCol), npre = (int)CS_REC(kRecNpre);
    const bool step_uniform = CS_REC(kRecStepUniform) != 0;
    const void *const a0_tab = CS_REC_PTR(const void *, kRecA0Tab), *const a1_tab = CS_REC_PTR(const void *, kRecA1Tab);
    const int a0_s0 = (int)CS_REC(kRecA0S0), a0_s2 = (int)CS_REC(kRecA0S2), a0_s3 = (int)CS_REC(kRecA0S3);
    const int a1_s = (int)CS_REC(kRecA1S1), a1_s2 = (int)CS_REC(kRecA1S2), a1_s3 = (int)CS_REC(kRecA1S3);
    int32_t *const status = CS_REC_PTR(int32_t *, kRecStatus);
    const int slab_begin = (int)CS_REC(kRecSlabBegin), halo_lo_p = (int)CS_REC(kRecHaloLo);
    const uint32_t js1 = CS_REC(kRecJs1), js2 = CS_REC(kRecJs2), js3 = CS_REC(kRecJs3);
    const int index_base = (int)CS_REC(kRecIndexBase), idx_bytes = (int)CS_REC(kRecIdxBytes);
#undef CS_REC_PTR
#undef CS_REC
    const int chunks = (n0 + LANES - 1) / LANES;
    // ---- hop 2: which column (see DColSweep::xcd_ig) - branch-free, one table look-up ------------------------------------
    int i2, i3, chunk, part;
    {
        unsigned item = (blockIdx.x >> 3) * 4u + (unsigned)wave;
        const bool win = cs_win != 0;
        const unsigned nfull = (unsigned)((GAX == 3) != win ? n2 : n3);      // the axis every XCD walks in full
        const unsigned per_part = cnt * (unsigned)chunks * nfull;
        if (item >= per_part * (unsigned)cs_split) return;                    // uniform over the wave
        part = (int)(item / per_part);                                        // which part of the column (outermost)
        item -= (unsigned)part * per_part;
        // win: the XCD owns window-axis indices, the group axis is walked in full, fastest; else it owns group-axis indices
        const unsigned fast = win ? (unsigned)(GAX == 3 ? n3 : n2) : cnt;      // extent of the fastest index
        const unsigned lo = item % fast, r = item / fast;
        chunk = (int)(r % (unsigned)chunks);
        const unsigned hi = r / (unsigned)chunks;
        const int look = xcd_tab[xcd * (unsigned)cs_xstride + (win ? hi : lo)];
        const int ig = win ? (int)lo : look, iw = win ? look : (int)hi;
        i2 = GAX == 3 ? iw : ig;
        i3 = GAX == 3 ? ig : iw;
    }
    int i0 = chunk * LANES + lane;
    bool valid = lane < LANES && i0 < n0;                   // this lane carries a state (see the one-load form below)
    if (!valid) i0 = n0 - 1;                                // halo / tail lanes: duplicate work, no store
    // ---- hop 3: everything that hangs on (i2, i3), asked for together: the plan's member slots (-> LDS) and this lane's axis-0
    // entry (vector loads, in flight across the scalar wait), then the plan's header words and the first axis-1 entry (scalar,
    // one batch by hand) ------------------------------------------------------------------------------------------------
    cptr<int32_t> pl = as_const<int32_t>(plan_base) + (size_t)(i2 + n2 * i3) * kCsPlanWords;
    f4 slot_v[(NG * MM * 2 + 63) / 64];
    {
        gptr<f4> src = as_global<f4>(plan_base + (size_t)(i2 + n2 * i3) * kCsPlanWords + kCsPI);
#pragma unroll
        for (int j = 0; j < (NG * MM * 2 + 63) / 64; ++j)
            if (lane + 64 * j < NG * MM * 2) slot_v[j] = src[lane + 64 * j];
    }
    TabEntry<T> e0v;
    {
        const int off = a0_s0 * i0 + a0_s2 * i2 + a0_s3 * i3;
        e0v.cell = as_global<TabEntry<T>>(a0_tab)[off].cell;
        e0v.t = as_global<TabEntry<T>>(a0_tab)[off].t;
    }
    const int a1_base = a1_s2 * i2 + a1_s3 * i3;
    cptr<TabEntry<T>> tab1 = as_const<TabEntry<T>>(a1_tab) + a1_base;
    const int i1b = (int)((int64_t)n1 * part / cs_split), i1e = (int)((int64_t)n1 * (part + 1) / cs_split);   // this wave's steps
    cptr<TabEntry<T>> tab1n = tab1 + i1b * a1_s;             // entry of the next step
    u16v hdr, cr0;
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    typedef uint32_t u8v __attribute__((ext_vector_type(8)));
    u2v ent1;
    u8v cr1;
    static_assert(1 + 2 * kCsGMax <= 16 && sizeof(TabEntry<T>) == 8, "plan header in one s_load_dwordx16, a table entry in one dwordx2");
    static_assert(offsetof(DColSweep, crec) == 192 && kCRecWords == 24, "the cost record follows the launch record");
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx2 %1, %5, 0x0\n\ts_load_dwordx16 %2, %6, 0xc0\n\ts_load_dwordx8 %3, %6, 0x100\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(hdr), "=&s"(ent1), "=&s"(cr0), "=&s"(cr1) : "s"(pl), "s"(tab1n), "s"(CS) : "memory");
#define CS_CREC(i) ((i) < 16 ? cr0[(i) & 15] : cr1[((i) - 16) & 7])
    int c1n = (int)ent1.x;
    T t1n = __uint_as_float(ent1.y);
#pragma unroll
    for (int j = 0; j < (NG * MM * 2 + 63) / 64; ++j)
        if (lane + 64 * j < NG * MM * 2) s_slots[wave][lane + 64 * j] = slot_v[j];
    const int ng = (int)hdr[0] >> 8;                         // groups of this column (<= NG)
    if ((hdr[0] & 1) && lane == 0) *status = 1;
    // ---- axis 0: the thread's own (cell, t) for the whole column ------------------------------------------
    uint32_t voff0;
    T t0;
    {
        const int c0 = e0v.cell;
        t0 = e0v.t;
        if (DPP) {
            // rel = cell - state index takes at most two adjacent values over the wave's states (verified on the host).
            // kb = the value most states share: lane L loads knot chunk start + L + kb, a state with rel == kb finds its
            // neighbours in its own lane and the next one.
            const int rel = c0 - i0;
            const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
            const int r0 = __builtin_amdgcn_readfirstlane(rel);                 // lane 0 is always a state
            const unsigned long long same = __builtin_amdgcn_ballot_w64(valid && rel == r0);
            int kb = r0;
            if (2 * __builtin_popcountll(same) < __builtin_popcountll(vm))
                kb = __builtin_amdgcn_readlane(rel, __builtin_ctzll(vm & ~same));
            const unsigned long long ex = __builtin_amdgcn_ballot_w64(valid && rel != kb);
            int knot = chunk * LANES + lane + kb;                               // the knot this lane loads
            if (ex != 0) {
                // ONE odd state (a cell clamped at the grid edge, typically): it moves to the spare lane pair behind the
                // halo lane - lane LANES + 1 takes the state and loads its cell's lower knot, lane LANES + 2 the upper one -
                // and its old lane stays as the loader its lower neighbour needs.  Every state lane is then regular.
                const int pe = __builtin_ctzll(ex);
                const int c0e = __builtin_amdgcn_readlane(c0, pe);
                const T t0e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t0), pe));
                if (lane == pe) valid = false;
                if (lane == LANES + 1) { i0 = chunk * LANES + pe; t0 = t0e; knot = c0e; valid = true; }
                if (lane == LANES + 2) knot = c0e + 1;
            }                                                                   // (the host admits no more than one)
            knot = knot < 0 ? 0 : (knot > n0 - 1 ? n0 - 1 : knot);              // clamped lanes are never referenced
            voff0 = (uint32_t)knot * (uint32_t)sizeof(TJ);
        } else {
            voff0 = (uint32_t)c0 * (uint32_t)sizeof(TJ);
        }
    }
    // byte offset of a corner row = (group's first row: scalar, from the plan) + (window knot w) * w_bytes + this lane's
    // axis-0 offset + the step's axis-1 row; every group has three valid window knots (the plan keeps windows inside the grid)
    uint32_t rog[NG];
    int used[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        rog[g] = hdr[1 + g];
        used[g] = (int)hdr[1 + kCsGMax + g];
    }
    // ---- cost: leading state-only terms that do not change along the column --------------------------------
    int si[D] = {i0, 0, i2, i3 + slab_begin};
    const int cjz[HJB_MAX_C] = {0, 0, 0};
    T gcol = (T)0;
    double gcol64 = 0.0;                          // C64: the column-constant state terms of the cost in double
    const bool rec_cost = !C64 && (int)CS_CREC(kCRecNCol) == npre_col;      // every column term is in the record (<= 3)
    if (rec_cost) {
        // their values requested together (the descriptors came with the plan header), then the ordered sum
        T x[3] = {(T)0, (T)0, (T)0};
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < npre_col) {
                const T *d = (const T *)(uintptr_t)((uint64_t)CS_CREC(kCRecTerm + 5 * j) | ((uint64_t)CS_CREC(kCRecTerm + 5 * j + 1) << 32));
                const int64_t off = (int64_t)(int)CS_CREC(kCRecTerm + 5 * j + 2) * si[0] + (int64_t)(int)CS_CREC(kCRecTerm + 5 * j + 3) * si[2] +
                                    (int64_t)(int)CS_CREC(kCRecTerm + 5 * j + 4) * si[3];
                x[j] = as_global<T>(d)[off];
            }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < npre_col) gcol = (j == 0) ? x[j] : (T)(gcol + x[j]);
    } else if constexpr (C64) {
        for (int k = 0; k < npre_col; ++k) {
            const double x = term_value<double, D>(P->cost64[k], si, cjz);
            gcol64 = (k == 0) ? x : gcol64 + x;
        }
    } else {
        for (int k = 0; k < npre_col; ++k) {
            const T x = term_value<T, D>(P->cost[k], si, cjz);
            gcol = (k == 0) ? x : (T)(gcol + x);
        }
    }
    // Scalar bases: (lower, upper) group row [x (lower, upper) axis-0 neighbour], and ONE 32-bit per-lane byte offset per
    // (group, window knot), advanced along the column: every gather is `global_load v, v_off, s[base]`, no 64-bit vector
    // arithmetic.  The bases are opaque to the compiler: the two loads of a corner pair must stay two instructions
    // (merged into one unaligned 8-byte load they are slower, measured on the row kernel).
    gptr<char> Jb00 = as_global<char>(Jn), Jb01 = Jb00, Jb11 = Jb00;
    gptr<char> Jb10 = as_global<char>(reinterpret_cast<const char *>(Jn) + g_bytes);
    asm volatile("" : "+s"(Jb10));
    if (!DPP) {
        Jb01 = as_global<char>(reinterpret_cast<const char *>(Jn) + sizeof(TJ));
        Jb11 = as_global<char>(reinterpret_cast<const char *>(Jn) + g_bytes + sizeof(TJ));
        asm volatile("" : "+s"(Jb01));
        asm volatile("" : "+s"(Jb11));
    }
    const uint32_t out_col = (uint32_t)i0 + js2 * (uint32_t)i2 + js3 * (uint32_t)(i3 + halo_lo_p);
    const uint32_t idx_col = (uint32_t)i0 + (uint32_t)n0 * (uint32_t)n1 * ((uint32_t)i2 + (uint32_t)n2 * (uint32_t)i3);
    // HJB_CS_DIRECT: per-lane byte offsets of the column's results (J < 4 GiB in this kernel: every gather offset is 32-bit too)
    const uint32_t out_off = out_col * (uint32_t)sizeof(TJ), idx_off = idx_col * (uint32_t)idx_bytes;
    typedef __attribute__((address_space(1))) char *gwptr;
    gwptr Jout_b = (gwptr)Jout, idx_b = (gwptr)idx_out;
    __builtin_amdgcn_wave_barrier();

    // the axis-0 lerp of one corner row from the value(s) a lane loaded
    auto xl = [&](T a, T b) -> T {
        if (DPP) return fma_t<T>(t0, lane_up_minus(a), a);      // a = the knot this lane loaded
        return fma_t<T>(t0, (T)(b - a), a);                      // a, b = the lower / upper neighbour
    };

    // ---- the column loop ------------------------------------------------------------------------------------------
    // Software pipeline in two halves of the group sequence, H0 = groups [0, NGH) and H1 = [NGH, NG): the rows of H1 are
    // requested before H0 is computed, the rows of the NEXT step's H0 before H1 is computed, so every load has half a
    // step of arithmetic (times the other waves of the SIMD) to land, and each half keeps its own registers - no
    // rotation.  Results go to LDS and are written out every kCsFlush steps: on gfx9 loads and stores share one counter
    // and complete out of order with each other, so a pending store makes every wait for a load a full drain.
    constexpr int NGH = (NG + 1) / 2;
    constexpr bool TWO = DPP && sizeof(TJ) == 4 && HJB_CS_ROLL2 != 0;         // two textual steps per loop trip
    constexpr bool ROLL2 = TWO && HJB_CS_ROLL2 != 2;                          // the copy-free roll (above); HJB_CS_ROLL2 = 2: two steps, round-4 registers (debugging)
    constexpr int NA = ROLL2 ? 2 : 1;                        // ROLL2: two sets of row pairs, each in turn gather destination / new A rows and old A rows
    f2 A[NA][NG][NW];
#pragma unroll
    for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int w = 0; w < NW; ++w) A[q][g][w] = f2{(T)0, (T)0};
    constexpr int LPK = DPP ? 2 : 4;                         // gathers per window knot of a group
    int ngs = ng;                                            // the group count as a value of the current step (below)
    uint32_t rlo[NG][2][NW], rhi[NG][2][NW];
    // Window knot 1 serves both slot pairs of a group, knot 0 only pair 0's slots, knot 2 only pair 1's: a knot no slot
    // uses is neither gathered nor lerped.  How many gathers each half issues is therefore a property of the column.
    const int pairmask = (1 << (MM / 2)) - 1;
    int nH0 = 0, nH1 = 0;
    // (A group the column does not have - fewer distinct cells at the grid's edge - has no slot in use and repeats group
    // 0's rows: its knot 1 is gathered all the same, which costs two L1 hits and saves a test per group and half.)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int n = LPK * (1 + ((used[g] & pairmask) != 0) + ((used[g] & (pairmask << (MM / 2))) != 0));
        if (g < NGH) nH0 += n; else nH1 += n;
    }
    // operands of a scalar jump (wait_gathers_n): pinned to scalar registers whatever pipe the compiler summed them on (under
    // scalar-register pressure it moves uniform arithmetic to the vector pipe, and an "s" asm operand then receives a VGPR -
    // which only the assembler rejects: tests/test_kernel_budget.py assembles the code object for that reason)
    nH0 = __builtin_amdgcn_readfirstlane(nH0);
    nH1 = __builtin_amdgcn_readfirstlane(nH1);
    // QQ (ROLL2): the set of row pairs that receives the rows - the one whose A rows have just been read for the last time
    auto load_groups = [&](int g0, int g1, uint32_t vrow, auto QQ) __attribute__((always_inline)) {
        constexpr int q = decltype(QQ)::value;
        uint32_t vb[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) vb[w] = voff0 + (vrow + (uint32_t)w * w_bytes);
#pragma unroll
        for (int g = g0; g < g1; ++g) {
            {
                int ug = used[g];
                asm volatile("" : "+s"(ug));                 // tested here, per step (see cs_group)
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (w == 0 && !(ug & 1)) continue;                          // a pair is in use iff its first slot is
                    if (w == 2 && !(ug & (1 << (MM / 2)))) continue;
                    const uint32_t o = vb[w] + rog[g];
                    if constexpr (ROLL2) {                   // into the registers of the old A row, by name: "+v"
                        uint32_t r0, r1;
#if HJB_CS_ROLL2 == 3
                        r0 = __float_as_uint(A[q][g][w].x), r1 = __float_as_uint(A[q][g][w].y);      // the destinations TIED to the old row's registers: faults on the GPU (round 5, not understood); kept for the record
                        asm volatile("global_load_dword %0, %1, %2" : "+v"(r0) : "v"(o), "s"(Jb00));
                        asm volatile("global_load_dword %0, %1, %2" : "+v"(r1) : "v"(o), "s"(Jb10));
#else
                        asm volatile("global_load_dword %0, %1, %2" : "=v"(r0) : "v"(o), "s"(Jb00));
                        asm volatile("global_load_dword %0, %1, %2" : "=v"(r1) : "v"(o), "s"(Jb10));
#endif
                        A[q][g][w] = f2{__uint_as_float(r0), __uint_as_float(r1)};
                        continue;
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        rlo[g][k][w] = gather_async<sizeof(TJ)>(o, k ? Jb10 : Jb00);
                        if (!DPP) rhi[g][k][w] = gather_async<sizeof(TJ)>(o, k ? Jb11 : Jb01);
                    }
                }
            }
        }
    };
    // wait until at most `younger` gathers (a multiple of LPK) are outstanding, then release groups [g0, g1) to the arithmetic
    auto await_groups = [&](int g0, int g1, int younger, auto QQ) __attribute__((always_inline)) {
        wait_gathers_n<LPK>(younger);
#pragma unroll
        for (int g = g0; g < g1; ++g) {
            if constexpr (ROLL2) {
                tie3(A[decltype(QQ)::value][g]);
            } else {
                tie6(rlo[g]);
                if (!DPP) tie6(rhi[g]);
            }
        }
    };
    static_assert(NG - (NG + 1) / 2 <= 3 && (NG + 1) / 2 <= 3, "await_groups counts up to nine younger window knots");
    T best, gstep, t1;
    double gstep64 = 0.0;
    int best_u;
    typedef __attribute__((address_space(3))) const f4 lds_f4;
    lds_f4 *slots = (lds_f4 *)&s_slots[wave][0];
    asm volatile("" : "+v"(slots));          // one address register for the column, not one re-made per read
    // The arithmetic (cs_group) is written on PAIRS (lower, upper group row) of one window knot: v_pk_add_f32 /
    // v_pk_fma_f32 are IEEE per component, and with the pair as the unit of data no value has to be moved between registers.
    auto compute_groups = [&](int g0, int g1, auto PP) __attribute__((always_inline)) {
        constexpr int pn = decltype(PP)::value, po = NA - 1 - pn;      // ROLL2: set pn holds the gathered rows and takes the new A rows, set po the old ones
#pragma unroll
        for (int g = g0; g < g1; ++g) {
            if (g < ngs) {
                const f2 t0p = {t0, t0};
                auto row0 = [&](int w) {
                    f2 l;
                    if constexpr (ROLL2) l = A[pn][g][w];
                    else l = f2{raw_to<T, TJ>(rlo[g][0][w]), raw_to<T, TJ>(rlo[g][1][w])};
                    const f2 d = DPP ? f2{lane_up_minus(l.x), lane_up_minus(l.y)}      // (next lane) - (own): one DPP subtraction each
                                     : f2{raw_to<T, TJ>(rhi[g][0][w]), raw_to<T, TJ>(rhi[g][1][w])} - l;
                    return __builtin_elementwise_fma(t0p, d, l);
                };
                cs_group<T, GAX, FASTCOST, decltype(row0), decltype(slots), C64>(used[g], g, row0, A[po][g], A[pn][g], t1, slots, gstep, ncu, npre, best, best_u, gstep64);
            }
        }
    };
    // the per-step cost term of the usual shape: its descriptor is read once (from the cost record), not once per step
    const bool one_su = step_uniform && npre - npre_col == 1;
    cptr<T> su_ptr;
    int su_s1 = 0;
    if (CS_CREC(kCRecHasSu) != 0 && one_su && !C64) {
        su_ptr = as_const<T>((const T *)(uintptr_t)((uint64_t)CS_CREC(kCRecSu) | ((uint64_t)CS_CREC(kCRecSu + 1) << 32))) +
                 ((int)CS_CREC(kCRecSu + 3) * i2 + (int)CS_CREC(kCRecSu + 4) * si[3]);
        su_s1 = (int)CS_CREC(kCRecSu + 2);
    } else {
        su_ptr = as_const<T>(P->cost[one_su ? npre_col : 0].data);
        if (one_su) {
            const DTerm &tm = P->cost[npre_col];
            su_ptr += tm.stride[2] * i2 + tm.stride[3] * si[3];
            su_s1 = tm.stride[1];
        }
    }
#undef CS_CREC
    int prev_c1 = -2;
    // Everything hipcc loaded for the set-up has landed before the loop starts: a value still "pending" at the loop
    // header would make its first use INSIDE the loop a `vmcnt(0)` on every step - a drain of the gathers in flight.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" : "+v"(gcol), "+v"(t0), "+v"(voff0));
    load_groups(0, NGH, (uint32_t)(c1n + 1) * s1_bytes, std::integral_constant<int, 0>{});      // prologue: H0 of step 0
    int slot = 0;                                            // LDS slot of this step's result (= i1 % kCsFlush)
    if constexpr (TWO) {
        for (int i1 = i1b; i1 < i1e; ++i1) {
            {
#define CS_PN 0
#include "kernels_colsweep_step.inc"
#undef CS_PN
            }
            if (++i1 >= i1e) break;
            {
#define CS_PN (NA - 1)
#include "kernels_colsweep_step.inc"
#undef CS_PN
            }
        }
    } else {
        for (int i1 = i1b; i1 < i1e; ++i1) {
#define CS_PN 0
#include "kernels_colsweep_step.inc"
#undef CS_PN
        }
    }
}

}  // namespace hjb
