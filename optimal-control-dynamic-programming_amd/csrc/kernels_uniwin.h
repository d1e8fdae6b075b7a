// kernels_uniwin.h - K15, variant 4 modes 7 / 8: the window kernel for grids whose RATE axes are shared by a whole chunk.
//
// K3's window modes (kernels_packed2.h, modes 5 / 6) serve Solver_attitude.run's shape (attitude-control/Solver_attitude.m:
// 261-300, 400-409, 413-506): D >= 4 state axes in the order (state-only axes ..., level-0 axis, level-1 axis, last axis),
// three control dims, the next value of axis D-3 moved by control dim 0, of axis D-2 by control dim 1, of the last axis by
// control dim 2.  On the attitude model those three are the body rates, and their next value depends on the RATES AND THE
// TORQUE ONLY (Solver_attitude.m:423-425) - never on the angles.  A 256-state chunk that lies inside ONE point of the three
// rate axes therefore shares, lane for lane,
//     * the last axis' query q_j = ql + b[j], its cell, its weight t_j, the control jc at which the cell changes;
//     * the (cell, t) entries of the level-0 and level-1 axes for every control level;
//     * every cost term over the controls.
// K3 keeps all of that per lane (48 bytes of weights per lane in LDS, entries and rows in vector registers, per-lane cell
// tests folded by ballots): 120 VGPRs and 40.5 KB of LDS per workgroup = four waves per SIMD.  This kernel keeps it per WAVE:
//     * a host-requested PLAN (k_uniwin_plan, once per problem): one 512-byte record per point of the rate axes with the
//       entries, the twelve (t_j, r_j), the cells and a flag that says whether the point has the usual shape;
//     * a wave reads its point's record as two coalesced dwords per lane and pulls what a trip needs out of them with
//       v_readlane: the weights and control costs of the inner sweep sit in SCALAR registers for the whole chunk and enter
//       the packed instructions as scalar operands - no LDS read, no wait, no ring of registers in the sweep;
//     * what is left per lane: the 27-entry window of the state-only axes (LDS, as in K3), two packed row sets of the
//       level-1 lerp, the running minima.  27.7 KB of LDS and <= 96 VGPRs: FIVE waves per SIMD.
// Chunks never straddle a point (the last chunk of a point is partly empty), so there is no per-lane fallback; a point whose
// sweep leaves the 2-cell windows (flagged in its record) takes a plain per-lane evaluation of every backup, and the host only
// chooses this kernel when such points are rare (hjbdp_setup.hip).
//
// The chunk walk is tiled for the L2: the points are cut into tiles of 8 x 4 x 4 and for one tile every
// angle chunk is visited in turn, a tile's 128 points back to back: the window slices those 128 workgroups touch at a time are
// (8+2) x (4+2) x (4+2) point-slices of ONE angle chunk's neighbourhood instead of 27 per point.  One generation of workgroups
// (grid = occupancy x CUs) walks it, workgroup b serving XCD b % 8, and the positions are CLAIMED from a per-XCD counter, not
// strided: that is what keeps the chunks in flight under one L2 neighbours (24^6 36.6 -> 30.0 ms, C3 3.96 -> 2.86 s per stage).
//
// Arithmetic per backup is the canonical order - the same operations in the same order as K3's window
// modes: bit-identical results.
#pragma once
#include <type_traits>
#include "hjbdp_dev.h"
#include "kernels_nested.h"
#include "kernels_packed.h"
#include "kernels_packed2.h"

namespace hjb {

constexpr int kUwRec = 128;        // dwords per plan record
constexpr int kUwMaxO = 16;        // most levels of an outer control dim (entries per axis in a record)
constexpr int kUwIn = 12;          // inner controls a record holds (11 or 12 run: six pair rows)
// plan record (dwords); first half = lane L's dword L of rec0, second half = rec1
constexpr int kUwA = 0;            // [0, 32)   level-0 axis: (cell, t) of o0 = 0 .. 15
constexpr int kUwB = 32;           // [32, 64)  level-1 axis: (cell, t) of o1 = 0 .. 15
constexpr int kUwT = 64;           // [64, 76)  t_j of the last axis, j = 0 .. 11 (a padding control: 0)
constexpr int kUwR = 76;           // [76, 88)  r_j, the inner control's cost (a padding control: +inf, never selected)
constexpr int kUwC = 88;           // [88, 100) last-axis cell of control j (GLOBAL cell index)
constexpr int kUwLc0 = 100;        // cell of control 0
constexpr int kUwLc1 = 101;        // cell after the first change (== lc0 when there is none)
constexpr int kUwJc = 102;         // first control in the second cell (m_in when there is none)
constexpr int kUwCa = 103;         // smallest level-0 cell over o0
constexpr int kUwCb = 104;         // smallest level-1 cell over o1
constexpr int kUwFlags = 105;      // bit 0: the point does not have the usual shape (slow path)

struct DUniwin {
    const int32_t *plan;           // [n_points][kUwRec]
    int32_t n_points;              // nA * nB * nC
    int32_t cpp;                   // chunks (of `block` states) per point
    int32_t inner;                 // states of the state-only axes (product of n[0 .. D-4])
    int32_t nA, nB, nC;            // points of the level-0 axis, the level-1 axis, OWNED planes of the last axis
    int32_t lA, lB, lC;            // log2 of the tile's extent along each
    int32_t ntA, ntB, ntC;         // tiles along each
    uint32_t tile_chunks;          // cpp << (lA + lB + lC): visiting positions per tile
    uint32_t n_v;                  // visiting positions in all
    int32_t cl1_per_o0;            // the level-1 cost term depends on o0 too (reloaded per o0 step)
    int32_t block;                 // states per chunk = threads per workgroup (256 or 64)
    uint32_t *counters;            // [8][16]: per XCD the next position of its walk (zeroed before every launch); null: static walk
};

// ---- the plan: one thread per point of the rate axes ---------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256)
k_uniwin_plan(const DParams *__restrict__ P, const DNested *__restrict__ N, int32_t *__restrict__ plan, int n_points, int nA, int nB,
              int32_t *__restrict__ n_slow) {
    constexpr int AX_A = D - 3, AX_B = D - 2;
    const DAxis &axl = P->axis[D - 1];
    const int m_in = N->m_in, m_o0 = N->m_o0, m_o1 = N->m_o1;
    for (int pt = blockIdx.x * blockDim.x + threadIdx.x; pt < n_points; pt += gridDim.x * blockDim.x) {
        int si[D];
#pragma unroll
        for (int a = 0; a < D; ++a) si[a] = 0;
        int r = pt;
        si[AX_A] = r % nA; r /= nA;
        si[AX_B] = r % nB; r /= nB;
        const int last_local = r;
        si[D - 1] = last_local + P->slab_begin;
        int32_t *rec = plan + (size_t)pt * kUwRec;
        for (int i = 0; i < kUwRec; ++i) rec[i] = 0;
        bool slow = false;
        // the last axis: q_j = ql + b[j] in the canonical order, exact cell, weight (kernels_packed2.h, "once per state")
        float ql = 0.f;
        for (int k = 0; k < axl.n_prefix; ++k) {
            const float x = term_value32<D>(axl.t[k], si);
            ql = (k == 0) ? x : ql + x;
        }
        const bool b_pure = N->in[0].lds_slot >= 0;
        int boff = 0;
        if (!b_pure) {
            const DTerm &bt = axl.t[N->ax_kin];
#pragma unroll
            for (int a = 0; a < D; ++a) boff += bt.stride[a] * si[a];
        }
        gptr<float> b_data = as_global<float>(N->in[0].data);
        const int b_stride = N->in[0].stride_in;
        gptr<float> kk = as_global<float>(axl.knots), rr = as_global<float>(axl.rdx);
        gptr<float> r_data = as_global<float>(N->in[kMaxInAx].data);
        const int r_stride = N->in[kMaxInAx].stride_in;
        int lc0 = 0, lc1 = 0, jc = m_in, nchange = 0, prev = 0;
        for (int j = 0; j < kUwIn; ++j) {
            if (j >= m_in) {
                rec[kUwT + j] = __float_as_int(0.f);
                rec[kUwR + j] = __float_as_int(INFINITY);
                rec[kUwC + j] = prev;
                continue;
            }
            const float q = ql + (b_pure ? b_data[j * b_stride] : b_data[boff + j * b_stride]);
            const int c = find_cell_g(kk, axl.n, q, axl.uniform, (float)axl.x0, (float)axl.inv_h);
            const float tj = (q - kk[c]) * rr[c];
            rec[kUwT + j] = __float_as_int(tj);
            rec[kUwR + j] = __float_as_int(r_data[j * r_stride]);
            rec[kUwC + j] = c;
            if (j == 0) {
                lc0 = lc1 = c;
            } else if (c != prev) {
                if (nchange == 0) { lc1 = c; jc = j; }
                ++nchange;
            }
            prev = c;
        }
        if (nchange > 1) slow = true;
        if (lc1 - lc0 > 1 || lc0 - lc1 > 1) slow = true;
        rec[kUwLc0] = lc0;
        rec[kUwLc1] = lc1;
        rec[kUwJc] = jc;
        // the level axes' entries, copied from their tables (k_prep_axis_table built them with the canonical arithmetic)
        int cmin[2] = {0x7fffffff, 0x7fffffff}, cmax[2] = {-1, -1};
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int a = w == 0 ? AX_A : AX_B;
            const int m = w == 0 ? m_o0 : m_o1;
            const int step = w == 0 ? N->at[a].c0 : N->at[a].c1;
            int off = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) off += N->at[a].sstride[d] * (d == D - 1 ? last_local : si[d]);
            gptr<i2v> tab = as_global<i2v>(N->at[a].tab);
            for (int o = 0; o < kUwMaxO; ++o) {
                const i2v e = tab[off + (o < m ? o : m - 1) * step];
                rec[(w == 0 ? kUwA : kUwB) + 2 * o] = e.x;
                rec[(w == 0 ? kUwA : kUwB) + 2 * o + 1] = e.y;
                if (o < m) {
                    cmin[w] = e.x < cmin[w] ? e.x : cmin[w];
                    cmax[w] = e.x > cmax[w] ? e.x : cmax[w];
                }
            }
            if (cmax[w] - cmin[w] > 1) slow = true;
        }
        rec[kUwCa] = cmin[0];
        rec[kUwCb] = cmin[1];
        rec[kUwFlags] = slow ? 1 : 0;
        if (slow && n_slow) atomicAdd(n_slow, 1);
    }
}

__device__ __forceinline__ int uw_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float uw_lanef(int v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(v, lane)); }


// Packed fp32 with the operand selects written out.  hipcc folds a {x, x} broadcast into op_sel only in some shapes; here the
// sweep's weights / control costs are SCALAR register pairs (t_2p, t_2p+1), (r_2p, r_2p+1) and the level-1 rows are packed as
// {row pair (0, 1), row pair (1, 2)}: left to the compiler every broadcast became a pair built with moves (64 more registers).
//   uw_fma_sb<H>(s, a, b)   = fma({s[H], s[H]}, a, b)            s: scalar pair
//   uw_add_sb<H>(a, s)      = a + {s[H], s[H]}
//   uw_fma_sel<SEL>(s, a, b) = fma(s, a<SEL>, b<SEL>)            SEL: 0 = {lo, lo}, 1 = {hi, hi}, 2 = {lo, hi}, 3 = {hi, lo}
template <int H>
__device__ __forceinline__ f2 uw_fma_sb(f2 s, f2 a, f2 b) {
    f2 d;
    if constexpr (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "s"(s), "v"(a), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "s"(s), "v"(a), "v"(b));
    return d;
}
template <int H>
__device__ __forceinline__ f2 uw_add_sb(f2 a, f2 s) {
    f2 d;
    if constexpr (H == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "s"(s));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "s"(s));
    return d;
}
template <int SEL>
__device__ __forceinline__ f2 uw_fma_sel(f2 s, f2 a, f2 b) {
    f2 d;
    if constexpr (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "s"(s), "v"(a), "v"(b));
    else if constexpr (SEL == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(d) : "s"(s), "v"(a), "v"(b));
    else if constexpr (SEL == 2) asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(s), "v"(a), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(d) : "s"(s), "v"(a), "v"(b));
    return d;
}
// min(min(a, b), c) as ONE instruction.  Written with __builtin_fminf the compiler quiets a possible signalling NaN of every operand
// it did not produce itself (v_max_f32 x, x, x) - the packed sums come out of asm blocks - two or three extra instructions per trip;
// v_min3_f32 of finite values is the same minimum (the contract covers finite cost-to-go values).
__device__ __forceinline__ float uw_min3(float a, float b, float c) {
#if HJB_UW_MIN3ASM
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return __builtin_fminf(__builtin_fminf(a, b), c);
#endif
}
__device__ __forceinline__ float uw_min2(float a, float b) {
#if HJB_UW_MIN3ASM
    float d;
    asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return __builtin_fminf(a, b);
#endif
}
// scalar (SMEM) reads of wave-uniform words: the constant address space tells the compiler that a uniform index is an s_load
typedef int i4v __attribute__((ext_vector_type(4)));
template <typename T> using cptr = const __attribute__((address_space(4))) T *;

__device__ __forceinline__ f2 uw_fma_sel_rt(int sel, f2 s, f2 a, f2 b) {
    switch (sel) {
        case 0: return uw_fma_sel<0>(s, a, b);
        case 1: return uw_fma_sel<1>(s, a, b);
        case 2: return uw_fma_sel<2>(s, a, b);
        default: return uw_fma_sel<3>(s, a, b);
    }
}

#ifndef HJB_UW_WAVES
#define HJB_UW_WAVES 5
#endif
// A/B switches (tools/mkab.sh ... -DHJB_UW_SMEM=1; measured in profiles/r06_k15_ab_round1.log, _round2.log): a trip's level-1
// entries and cost pair by scalar loads instead of v_readlane (-7 % vector instructions, +50 % scalar: 1 - 4 % SLOWER); the level-0
// rows as packed lerps and the sweep's minima as hand-written v_min3 / v_min (fewer instructions, equal time within 1 %): all off.
// What pays is HJB_UW_NEST: -8 % (the scalar stream of a trip - a 25-way compare tree - was as long as half its vector stream)
#ifndef HJB_UW_SMEM
#define HJB_UW_SMEM 0
#endif
#ifndef HJB_UW_PKROWS
#define HJB_UW_PKROWS 0
#endif
#ifndef HJB_UW_MIN3ASM
#define HJB_UW_MIN3ASM 0
#endif
// 1: the whole (o0, o1) loop nest is instantiated per sweep shape behind ONE jump per chunk (the shape is the point's: it does not
// change inside a chunk); 0: one jump per trip / single step inside a common nest
#ifndef HJB_UW_NEST
#define HJB_UW_NEST 1
#endif
// > 0 (with HJB_UW_NEST): a second instantiation of every nest for THIS level-1 control count with the trips of an o0 step written
// out (lane reads at constant lanes, no loop bookkeeping) - the attitude grids' 11; other counts take the loop
#ifndef HJB_UW_UNROLL_O1
#define HJB_UW_UNROLL_O1 0
#endif
// 1: the running minimum is kept per TRIP (min(mA, mB) against the best so far: 5 vector instructions), and which of the trip's two
// steps - and which control - reached it is found when the winning trip is evaluated once more per state; 0: per step (9 of a trip's
// 73 vector instructions).  Same result: the first (step, control) in sweep order whose total equals the minimum.  Built and
// measured (profiles/r06_k15_triptrack.log): bit-exact on the whole K15 suite, 24^6 30.27 -> 30.09 ms, C3 equal - off.
#ifndef HJB_UW_TRIPTRACK
#define HJB_UW_TRIPTRACK 0
#endif

// BLOCK = states per workgroup = per chunk: 256 (four waves; 27.7 KB of LDS: five workgroups = 20 waves per CU) or 64 (ONE wave per
// workgroup: 7 KB of LDS, 23 workgroups per CU - the LDS is handed out in finer pieces - at <= 80 VGPRs: 5.75 waves per SIMD)
template <typename TJ, int D, bool QMODEL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BLOCK == 64 ? 6 : HJB_UW_WAVES)))
k_backup_uniwin(const DParams *__restrict__ P, const DNested *__restrict__ N, const DUniwin *__restrict__ U,
                const TJ *__restrict__ Jn, TJ *__restrict__ Jout, void *__restrict__ idx_out) {
    static_assert(D >= 4, "window kernel: at least one state-only axis");
    constexpr int NP = D - 3, AX_A = D - 3, AX_B = D - 2, DM = D - 1;
    constexpr int CL0 = HJB_MAX_D, CL1 = HJB_MAX_D + 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // LDS: the per-lane window W[(ra * 3 + rb) * 3 + q][lane] | the level cost terms' control tables
    float *my_w = reinterpret_cast<float *>(smem_raw) + threadIdx.x;
    float *s_ot = reinterpret_cast<float *>(smem_raw) + 27 * BLOCK;
#pragma unroll
    for (int i = CL0; i <= CL1; ++i) {
        const auto &t = N->ot[i];
        if (t.present)
            for (int e = threadIdx.x; e < t.lds_len; e += blockDim.x) s_ot[t.lds_off + e] = as_global<float>(t.data)[e];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int m_in = N->m_in, m_o0 = N->m_o0, m_o1 = N->m_o1;
    const int plane0 = P->plane0, nplanes = P->nplanes;
    int js[D];
#pragma unroll
    for (int a = 0; a < D; ++a) js[a] = a == 0 ? 1 : (int)P->jstride[a];
    const int nA = U->nA, nB = U->nB;
    const int inner = U->inner;
    const bool cl0_present = N->ot[CL0].present, cl1_present = N->ot[CL1].present;
    const bool cl0_first = N->ot[CL0].first, cl1_first = N->ot[CL1].first;
    // the level cost terms, lane o holding the term of control level o: cl0 once per kernel, cl1 once per kernel unless it
    // depends on o0 as well (then once per o0 step).  An absent level-1 term reads -0: g + (-0) == g bit for bit.
    float cl0v = 0.f, cl1v = -0.0f;
    if (cl0_present) cl0v = s_ot[N->ot[CL0].lds_off + (lane < m_o0 ? lane : 0) * N->ot[CL0].c0];
    const int cl1_off = N->ot[CL1].lds_off, cl1_c0 = N->ot[CL1].c0, cl1_c1 = N->ot[CL1].c1;
    const bool cl1_per_o0 = U->cl1_per_o0 != 0;
    cptr<float> cl1_base = (cptr<float>)N->ot[CL1].data;
    if (cl1_present && !cl1_per_o0) cl1v = s_ot[cl1_off + (lane < m_o1 ? lane : 0) * cl1_c1];

    gptr<i2v> atab[NP];
#pragma unroll
    for (int a = 0; a < NP; ++a) atab[a] = as_global<i2v>(N->at[a].tab);

    // visiting order (see the header): workgroup b serves XCD b % 8 with the (b % 8)-th contiguous eighth of every grid-sized span.
    // The positions are CLAIMED, not strided: a workgroup that is done takes the next position of its XCD's walk from a counter
    // (one atomic per 256-state chunk).  With a fixed stride the workgroups of an XCD drift apart over the tens of thousands of
    // chunks each of them sweeps at 51^6, the chunks in flight stop being neighbours and the tile's window slices are fetched
    // once per chunk instead of once per tile (C3: 7.6 TB of fabric traffic per stage against 0.18 TB algorithmic).
    __shared__ unsigned int s_claim[2];
    const unsigned int xcd = blockIdx.x & 7u, per_xcd = gridDim.x >> 3;          // (the grid is a multiple of 8: hjbdp_setup.hip)
    uint32_t *claim = U->counters ? U->counters + 16 * xcd : nullptr;
    unsigned int first_v = xcd * per_xcd + (blockIdx.x >> 3), turn = 0;
    const unsigned int n_v = U->n_v, tile_chunks = U->tile_chunks;
    const int lA = U->lA, lB = U->lB, lC = U->lC;
    const unsigned int ntA = (unsigned int)U->ntA, ntB = (unsigned int)U->ntB;
    for (unsigned int v = first_v;; v += gridDim.x) {
        if (claim) {
            if (threadIdx.x == 0) {
                const unsigned int k = atomicAdd(claim, 1u), span = k / per_xcd;
                s_claim[turn & 1u] = span * gridDim.x + xcd * per_xcd + (k - span * per_xcd);
            }
            __syncthreads();                       // (one barrier per chunk: the slot of turn t is rewritten at turn t + 2, behind barrier t + 1)
            v = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_claim[turn & 1u]);
            ++turn;
        }
        if (v >= n_v) break;
        // ---- which chunk (all scalar) ---------------------------------------------------------------------------------------
        const unsigned int tile = v / tile_chunks, rem = v - tile * tile_chunks;
        const unsigned int ci = rem >> (lA + lB + lC), p = rem & ((1u << (lA + lB + lC)) - 1u);
        const unsigned int tB_ = tile / ntA, tA_ = tile - tB_ * ntA, tC_ = tB_ / ntB, tBB = tB_ - tC_ * ntB;
        const int ia = (int)((tA_ << lA) | (p & ((1u << lA) - 1u)));
        const int ib = (int)((tBB << lB) | ((p >> lA) & ((1u << lB) - 1u)));
        const int ic = (int)((tC_ << lC) | (p >> (lA + lB)));
        if (ia >= nA || ib >= nB || ic >= U->nC) continue;
        const int pt = ia + nA * (ib + nB * ic);
        gptr<int> rec = as_global<int>(U->plan) + (size_t)pt * kUwRec;
        const int rec0 = rec[lane], rec1 = rec[64 + lane];
        // ... and the same record through the scalar cache: what a trip needs of it (the level-1 entries of its two steps) is ONE
        // s_load_dwordx4 - no vector instruction at all (a v_readlane is one)
        cptr<i4v> recB = (cptr<i4v>)(const void *)(U->plan + (size_t)pt * kUwRec + kUwB);

        // ---- this lane's state -------------------------------------------------------------------------------------------------
        int ii = (int)ci * BLOCK + (int)threadIdx.x;
        const bool valid = ii < inner;
        if (!valid) ii = inner - 1;                // harmless duplicate work, store skipped
        int si[D];
        {
            uint32_t r = (uint32_t)ii;
#pragma unroll
            for (int a = 0; a < NP; ++a) {
                const uint32_t q = udiv_gm(r, N->div_m[a], N->div_s[a]);
                si[a] = (int)(r - q * (uint32_t)P->n[a]);
                r = q;
            }
        }
        si[AX_A] = ia;
        si[AX_B] = ib;
        si[D - 1] = ic + P->slab_begin;
        float gpre = 0.f;
        for (int k = 0; k < P->n_cost_prefix; ++k) {
            const float x = term_value32<D>(P->cost[k], si);
            gpre = (k == 0) ? x : gpre + x;
        }
        int cell[NP];
        float tw[DM];
        if constexpr (QMODEL) {
            float qn[3];
            model_quat_next(P, si, as_global<float>(P->axis[3].knots)[si[3]], as_global<float>(P->axis[4].knots)[si[4]],
                            as_global<float>(P->axis[5].knots)[si[5]], qn);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const DAxis &ax = P->axis[a];
                gptr<float> kk = as_global<float>(ax.knots);
                const int c = find_cell_g(kk, ax.n, qn[a], ax.uniform, (float)ax.x0, (float)ax.inv_h);
                cell[a] = c;
                tw[a] = (qn[a] - kk[c]) * as_global<float>(ax.rdx)[c];
            }
        } else {
#pragma unroll
            for (int a = 0; a < NP; ++a) {
                int off = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) off += N->at[a].sstride[d] * (d == D - 1 ? ic : si[d]);
                const i2v e = atab[a][off];
                cell[a] = e.x;
                tw[a] = __int_as_float(e.y);
            }
        }
        int pbase = 0;
#pragma unroll
        for (int a = 0; a < NP; ++a) pbase += js[a] * cell[a];

        // ---- the point's scalars ----------------------------------------------------------------------------------------------
        const int flags = uw_lane(rec1, kUwFlags - 64);
        int lc0 = uw_lane(rec1, kUwLc0 - 64) - plane0, lc1 = uw_lane(rec1, kUwLc1 - 64) - plane0;
        const int jc = uw_lane(rec1, kUwJc - 64);
        const int cAmin = uw_lane(rec1, kUwCa - 64), cBmin = uw_lane(rec1, kUwCb - 64);
        bool slow = (flags & 1) != 0;
        if (lc0 < 0 || lc0 + 1 >= nplanes || lc1 < 0 || lc1 + 1 >= nplanes) slow = true;     // (the slow path reports it)

        float best = 0.f;
        int label = 0;
        if (!slow) {
            const int pl0 = lc0 < lc1 ? lc0 : lc1;
            const int qa = lc0 - pl0, qb = lc1 - pl0;
            // ---- the window: the state-only axes contracted over 3 x 3 rows x 3 planes (kernels_packed2.h, W3P) -------------------
            {
                int planes[3];
                planes[0] = pl0;
                planes[1] = pl0 + 1;
                planes[2] = pl0 + 2 < nplanes ? pl0 + 2 : nplanes - 1;
#pragma unroll 1
                for (int ra = 0; ra < 3; ++ra) {
                    const int rowA = cAmin + ra < nA ? cAmin + ra : nA - 1;
#pragma unroll
                    for (int rb = 0; rb < 3; ++rb) {
                        const int rowB = cBmin + rb < nB ? cBmin + rb : nB - 1;
                        const int off = pbase + js[AX_A] * rowA + js[AX_B] * rowB;
                        float vv[3][1 << NP];
#pragma unroll
                        for (int q = 0; q < 3; ++q) gather_corners<TJ, NP>(Jn + (int64_t)js[D - 1] * planes[q], off, js, vv[q]);
#pragma unroll
                        for (int q = 0; q < 3; ++q) my_w[((ra * 3 + rb) * 3 + q) * BLOCK] = contract_corners<NP>(vv[q], tw);
                    }
                }
            }
            // the inner sweep's weights and control costs: scalar registers for the whole chunk
            f2 tp[kUwIn / 2], rp[kUwIn / 2];             // pair row p = controls (2p, 2p + 1)
#pragma unroll
            for (int k = 0; k < kUwIn / 2; ++k) {
                tp[k] = (f2){uw_lanef(rec1, kUwT - 64 + 2 * k), uw_lanef(rec1, kUwT - 64 + 2 * k + 1)};
                rp[k] = (f2){uw_lanef(rec1, kUwR - 64 + 2 * k), uw_lanef(rec1, kUwR - 64 + 2 * k + 1)};
            }
            const int nfull = m_in >> 1;
            const int sel = nfull * 64 + ((jc >> 1) < nfull ? (jc >> 1) : nfull) * 4 + (((jc & 1) && jc < m_in) ? 2 : 0) +
                            (((m_in & 1) && jc == m_in - 1) ? 1 : 0);
            int best_uo = 0;
            auto nest = [&](auto NFn, auto PAn, auto STn, auto LSn, auto MO1n) __attribute__((always_inline)) {
                (void)NFn; (void)PAn; (void)STn; (void)LSn;
                constexpr int MO1 = decltype(MO1n)::value;                 // the level-1 control count when it is a constant of this copy (else 0)
                const int mo1 = MO1 > 0 ? MO1 : m_o1;
            for (int o0 = 0; o0 < m_o0; ++o0) {
                // ---- level 0: the axis' entry, the three level-1 rows x four planes it lerps, as two packed row sets --------------
                const int cA = uw_lane(rec0, kUwA + 2 * o0);
                const float tA = uw_lanef(rec0, kUwA + 2 * o0 + 1);
                const f2 tAp = {tA, tA};
                const float *w0 = my_w + ((cA - cAmin) * 9) * BLOCK;
                // R0[q] = {F[0][q], F[1][q]}, RD[q] = {F[1][q] - F[0][q], F[2][q] - F[1][q]}: a step on level-1 row pair (0, 1) takes
                // the first halves, on (1, 2) the second - picked by the packed instructions' operand selects, nothing is re-formed
                f2 R0[4], RD[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int wq = (q < 2 ? qa : qb) + (q & 1);
#if !HJB_UW_PKROWS
                    float F[3];
#pragma unroll
                    for (int rb = 0; rb < 3; ++rb) {
                        const float f0 = w0[(rb * 3 + wq) * BLOCK];
                        const float f1 = w0[(9 + rb * 3 + wq) * BLOCK];
                        F[rb] = __builtin_fmaf(tA, f1 - f0, f0);
                    }
                    R0[q] = (f2){F[0], F[1]};
                    RD[q] = (f2){F[1] - F[0], F[2] - F[1]};
                    (void)tAp;
                    continue;
#endif
                    // {F[0], F[1]} and {F[1], F[2]} as packed lerps of the level-0 axis (F[rb] = fma(tA, f1 - f0, f0) element for element)
                    const f2 a01 = {w0[(0 * 3 + wq) * BLOCK], w0[(1 * 3 + wq) * BLOCK]}, a12 = {w0[(1 * 3 + wq) * BLOCK], w0[(2 * 3 + wq) * BLOCK]};
                    const f2 b01 = {w0[(9 + 0 * 3 + wq) * BLOCK], w0[(9 + 1 * 3 + wq) * BLOCK]}, b12 = {w0[(9 + 1 * 3 + wq) * BLOCK], w0[(9 + 2 * 3 + wq) * BLOCK]};
                    const f2 F01 = uw_fma_sb<0>(tAp, b01 - a01, a01), F12 = uw_fma_sb<0>(tAp, b12 - a12, a12);
                    R0[q] = F01;
                    RD[q] = F12 - F01;
                }
                float go0 = gpre;
                if (cl0_present) {
                    const float x = uw_lanef(__float_as_int(cl0v), o0);
                    go0 = cl0_first ? x : go0 + x;
                }
                const float go0_l1 = cl1_first ? -0.0f : go0;
                if (cl1_present && cl1_per_o0) cl1v = s_ot[cl1_off + o0 * cl1_c0 + (lane < m_o1 ? lane : 0) * cl1_c1];
                const int cl1i = __float_as_int(cl1v);
                cptr<float> cl1g = cl1_base + o0 * cl1_c0;                      // the level-1 cost term of this o0 step, by scalar loads
                const int uo0 = o0 * m_o1;
                int o1 = 0;
                i4v eb_nx = recB[0];
                f2 c2_nx = {-0.0f, -0.0f};                                          // (an absent term: g + (-0) == g bit for bit)
                if (cl1_present && m_o1 > 1) c2_nx = (f2){cl1g[0], cl1g[cl1_c1]};
                // ---- two (o0, o1) steps per trip, the two STEPS in the halves of every packed instruction -------------------------
#if HJB_UW_UNROLL_O1 > 0
#pragma unroll
#else
#pragma unroll 1
#endif
                for (; o1 + 1 < mo1; o1 += 2) {
#if HJB_UW_SMEM
                    const i4v eb = eb_nx;                                           // (cell, t) of steps o1 and o1 + 1
                    const f2 c2 = c2_nx;                                            // the level-1 cost term of the two steps
                    {   // ... and the next trip's, requested now: a trip never waits for the scalar cache
                        const int last = (m_o1 >> 1) - 1, nx0 = (o1 >> 1) + 1, nx = nx0 < last ? nx0 : (last > 0 ? last : 0);
                        eb_nx = recB[nx];
                        if (cl1_present) c2_nx = (f2){cl1g[2 * nx * cl1_c1], cl1g[(2 * nx + 1) * cl1_c1]};
                    }
#else
                    const i4v eb = {uw_lane(rec0, kUwB + 2 * o1), uw_lane(rec0, kUwB + 2 * o1 + 1), uw_lane(rec0, kUwB + 2 * o1 + 2),
                                    uw_lane(rec0, kUwB + 2 * o1 + 3)};
                    const f2 c2 = {uw_lanef(cl1i, o1), uw_lanef(cl1i, o1 + 1)};
                    (void)eb_nx; (void)c2_nx; (void)cl1g;
#endif
                    const int rA = eb.x - cBmin, rB = eb.z - cBmin;
                    const f2 t2 = {__int_as_float(eb.y), __int_as_float(eb.w)};
                    f2 X2[4];
                    switch (rA * 2 + rB) {                                          // (scalar: which level-1 row pair each step sits on)
                        case 0:
#pragma unroll
                            for (int q = 0; q < 4; ++q) X2[q] = uw_fma_sel<0>(t2, RD[q], R0[q]);
                            break;
                        case 3:
#pragma unroll
                            for (int q = 0; q < 4; ++q) X2[q] = uw_fma_sel<1>(t2, RD[q], R0[q]);
                            break;
                        case 1:
#pragma unroll
                            for (int q = 0; q < 4; ++q) X2[q] = uw_fma_sel<2>(t2, RD[q], R0[q]);
                            break;
                        default:
#pragma unroll
                            for (int q = 0; q < 4; ++q) X2[q] = uw_fma_sel<3>(t2, RD[q], R0[q]);
                            break;
                    }
                    const f2 Ea = X2[0], Da = X2[1] - X2[0], Eb = X2[2], Db = X2[3] - X2[2];
                    const f2 g2 = (f2){go0_l1, go0_l1} + c2;
                    float mA = INFINITY, mB = INFINITY;
                    auto pairs_fixed = [&](auto NFc, auto PAc, auto STc, auto LSc) __attribute__((always_inline)) {
                        constexpr int NF = decltype(NFc)::value, PA = decltype(PAc)::value;
                        constexpr bool ST = decltype(STc)::value, LS = decltype(LSc)::value;
#pragma unroll
                        for (int q = 0; q < NF; ++q) {
                            const f2 Ex = (q < PA || (ST && q == PA)) ? Ea : Eb, Dx = (q < PA || (ST && q == PA)) ? Da : Db;
                            const f2 Ey = q < PA ? Ea : Eb, Dy = q < PA ? Da : Db;
                            const f2 totx = uw_add_sb<0>(g2, rp[q]) + uw_fma_sb<0>(tp[q], Dx, Ex);
                            const f2 toty = uw_add_sb<1>(g2, rp[q]) + uw_fma_sb<1>(tp[q], Dy, Ey);
                            mA = uw_min3(mA, totx.x, toty.x);
                            mB = uw_min3(mB, totx.y, toty.y);
                        }
                        if constexpr (NF < kUwIn / 2) {             // 11 controls: the last one stands alone
                            constexpr bool second = PA < NF || LS;
                            const f2 El = second ? Eb : Ea, Dl = second ? Db : Da;
                            const f2 totx = uw_add_sb<0>(g2, rp[NF]) + uw_fma_sb<0>(tp[NF], Dl, El);
                            mA = uw_min2(mA, totx.x);
                            mB = uw_min2(mB, totx.y);
                        }
                    };
#if HJB_UW_NEST
                    pairs_fixed(NFn, PAn, STn, LSn);
#else
#define HJB_PF(NF, PA, ST, LS) case (NF) * 64 + (PA) * 4 + (ST) * 2 + (LS): pairs_fixed(std::integral_constant<int, NF>{}, std::integral_constant<int, PA>{}, \
                                       std::integral_constant<bool, (ST) != 0>{}, std::integral_constant<bool, (LS) != 0>{}); break;
                    switch (sel) {
                        HJB_PF(5, 0, 0, 0) HJB_PF(5, 0, 1, 0) HJB_PF(5, 1, 0, 0) HJB_PF(5, 1, 1, 0) HJB_PF(5, 2, 0, 0) HJB_PF(5, 2, 1, 0)
                        HJB_PF(5, 3, 0, 0) HJB_PF(5, 3, 1, 0) HJB_PF(5, 4, 0, 0) HJB_PF(5, 4, 1, 0) HJB_PF(5, 5, 0, 0) HJB_PF(5, 5, 0, 1)
                        HJB_PF(6, 0, 0, 0) HJB_PF(6, 0, 1, 0) HJB_PF(6, 1, 0, 0) HJB_PF(6, 1, 1, 0) HJB_PF(6, 2, 0, 0) HJB_PF(6, 2, 1, 0)
                        HJB_PF(6, 3, 0, 0) HJB_PF(6, 3, 1, 0) HJB_PF(6, 4, 0, 0) HJB_PF(6, 4, 1, 0) HJB_PF(6, 5, 0, 0) HJB_PF(6, 5, 1, 0)
                        HJB_PF(6, 6, 0, 0)
                        default: __builtin_unreachable();
                    }
#undef HJB_PF
#endif
                    const int uo = uo0 + o1;
#if HJB_UW_TRIPTRACK
                    const float mAB = uw_min2(mA, mB);
                    if (uo == 0 || mAB < best) { best = mAB; best_uo = uo; }      // (the trip's FIRST step; see the label below)
#else
                    if (uo == 0 || mA < best) { best = mA; best_uo = uo; }
                    if (mB < best) { best = mB; best_uo = uo + 1; }
#endif
                }
                // ---- the last step of an odd count: one step, two CONTROLS in the halves -----------------------------------------
                if (o1 < mo1) {
                    const int rA = uw_lane(rec0, kUwB + 2 * o1) - cBmin;
                    const float t1 = uw_lanef(rec0, kUwB + 2 * o1 + 1);
                    float X[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) X[q] = rA == 0 ? __builtin_fmaf(t1, RD[q].x, R0[q].x) : __builtin_fmaf(t1, RD[q].y, R0[q].y);
                    const f2 e2 = {X[0], X[2]}, d2 = {X[1] - X[0], X[3] - X[2]};          // {first cell, second cell}
                    const float go = go0_l1 + uw_lanef(cl1i, o1);
                    const f2 go2 = {go, go};
                    float ibest = INFINITY;
                    auto step_fixed = [&](auto PAc, auto STc) __attribute__((always_inline)) {
                        constexpr int PA = decltype(PAc)::value;
                        constexpr bool ST = decltype(STc)::value;
#pragma unroll
                        for (int q = 0; q < kUwIn / 2; ++q) {
                            const int sel2 = q < PA ? 0 : ((ST && q == PA) ? 2 : 1);       // (a constant once unrolled)
                            const f2 tot = (go2 + rp[q]) + uw_fma_sel_rt(sel2, tp[q], d2, e2);
                            ibest = uw_min3(ibest, tot.x, tot.y);
                        }
                    };
#if HJB_UW_NEST
                    {
                        constexpr int NFq = decltype(NFn)::value, PAq = decltype(PAn)::value;
                        constexpr bool STq = decltype(STn)::value, LSq = decltype(LSn)::value;
                        constexpr int JJ = STq ? 2 * PAq + 1 : (LSq ? 2 * NFq : (PAq < NFq ? 2 * PAq : kUwIn));     // the cell change's control
                        step_fixed(std::integral_constant<int, JJ / 2>{}, std::integral_constant<bool, (JJ & 1) != 0>{});
                    }
#else
#define HJB_SF(J) case (J): step_fixed(std::integral_constant<int, (J) / 2>{}, std::integral_constant<bool, ((J) & 1) != 0>{}); break;
                    switch (jc < kUwIn ? jc : kUwIn) {
                        HJB_SF(1) HJB_SF(2) HJB_SF(3) HJB_SF(4) HJB_SF(5) HJB_SF(6) HJB_SF(7) HJB_SF(8) HJB_SF(9) HJB_SF(10) HJB_SF(11) HJB_SF(12)
                        default: __builtin_unreachable();          // 1 <= jc: control 0 opens the first cell
                    }
#undef HJB_SF
#endif
                    const int uo = uo0 + o1;
                    if (uo == 0 || ibest < best) { best = ibest; best_uo = uo; }
                }
            }
            };
#if HJB_UW_NEST
#define HJB_NS(NF, PA, ST, LS) case (NF) * 64 + (PA) * 4 + (ST) * 2 + (LS): nest(std::integral_constant<int, NF>{}, std::integral_constant<int, PA>{}, \
                                       std::integral_constant<bool, (ST) != 0>{}, std::integral_constant<bool, (LS) != 0>{}, HJB_NS_MO1{}); break;
#if HJB_UW_UNROLL_O1 > 0
#define HJB_NS_MO1 std::integral_constant<int, HJB_UW_UNROLL_O1>
            if (m_o1 == HJB_UW_UNROLL_O1) {
                switch (sel) {
                    HJB_NS(5, 0, 1, 0) HJB_NS(5, 1, 0, 0) HJB_NS(5, 1, 1, 0) HJB_NS(5, 2, 0, 0) HJB_NS(5, 2, 1, 0)
                    HJB_NS(5, 3, 0, 0) HJB_NS(5, 3, 1, 0) HJB_NS(5, 4, 0, 0) HJB_NS(5, 4, 1, 0) HJB_NS(5, 5, 0, 0) HJB_NS(5, 5, 0, 1)
                    HJB_NS(6, 0, 1, 0) HJB_NS(6, 1, 0, 0) HJB_NS(6, 1, 1, 0) HJB_NS(6, 2, 0, 0) HJB_NS(6, 2, 1, 0)
                    HJB_NS(6, 3, 0, 0) HJB_NS(6, 3, 1, 0) HJB_NS(6, 4, 0, 0) HJB_NS(6, 4, 1, 0) HJB_NS(6, 5, 0, 0) HJB_NS(6, 5, 1, 0)
                    HJB_NS(6, 6, 0, 0)
                    default: __builtin_unreachable();
                }
            } else
#undef HJB_NS_MO1
#endif
#define HJB_NS_MO1 std::integral_constant<int, 0>
            switch (sel) {
                HJB_NS(5, 0, 1, 0) HJB_NS(5, 1, 0, 0) HJB_NS(5, 1, 1, 0) HJB_NS(5, 2, 0, 0) HJB_NS(5, 2, 1, 0)
                HJB_NS(5, 3, 0, 0) HJB_NS(5, 3, 1, 0) HJB_NS(5, 4, 0, 0) HJB_NS(5, 4, 1, 0) HJB_NS(5, 5, 0, 0) HJB_NS(5, 5, 0, 1)
                HJB_NS(6, 0, 1, 0) HJB_NS(6, 1, 0, 0) HJB_NS(6, 1, 1, 0) HJB_NS(6, 2, 0, 0) HJB_NS(6, 2, 1, 0)
                HJB_NS(6, 3, 0, 0) HJB_NS(6, 3, 1, 0) HJB_NS(6, 4, 0, 0) HJB_NS(6, 4, 1, 0) HJB_NS(6, 5, 0, 0) HJB_NS(6, 5, 1, 0)
                HJB_NS(6, 6, 0, 0)
                default: __builtin_unreachable();          // 1 <= jc: control 0 opens the first cell
            }
#undef HJB_NS
#undef HJB_NS_MO1
#else
            nest(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::false_type{}, std::false_type{}, std::integral_constant<int, 0>{});
#endif
            // ---- which inner control: the winning step's controls once more, in order (first-minimum rule) ----------------------
            int best_j = 0;
            {
                const int o0 = (int)udiv_gm((uint32_t)best_uo, N->div_m_o1, N->div_s_o1), o1f = best_uo - o0 * m_o1;
                const int cA = __builtin_amdgcn_ds_bpermute((kUwA + 2 * o0) * 4, rec0);
                const float tA = __int_as_float(__builtin_amdgcn_ds_bpermute((kUwA + 2 * o0 + 1) * 4, rec0));
                float g0 = gpre;
                if (cl0_present) {
                    const float x = s_ot[N->ot[CL0].lds_off + o0 * N->ot[CL0].c0];
                    g0 = cl0_first ? x : g0 + x;
                }
                const int ra = cA - cAmin;
                bool found = false;
                int best_o1 = o1f;
                // HJB_UW_TRIPTRACK: best_uo names the winning trip's first step; its second step (if it has one: trips start at even
                // o1, the last step of an odd count stands alone) is searched after it - first (step, control) in sweep order
#pragma unroll
                for (int sx = 0; sx < (HJB_UW_TRIPTRACK ? 2 : 1); ++sx) {
                    const bool has = sx == 0 || o1f + 1 < m_o1;
                    const int o1 = has ? o1f + sx : o1f;                              // (lane reads stay inside the record)
                    const int cB = __builtin_amdgcn_ds_bpermute((kUwB + 2 * o1) * 4, rec0);      // (every lane takes part: a lane
                    const float tB = __int_as_float(__builtin_amdgcn_ds_bpermute((kUwB + 2 * o1 + 1) * 4, rec0));   // read of an idle lane is 0)
                    float g = g0;
                    if (cl1_present) {
                        const float x = s_ot[cl1_off + o0 * cl1_c0 + o1 * cl1_c1];
                        g = cl1_first ? x : g + x;
                    }
                    const int rb = cB - cBmin;
                    float xe[2], xd[2];
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const int q0 = w == 0 ? qa : qb;
                        float X[2];
#pragma unroll
                        for (int dq = 0; dq < 2; ++dq) {
                            float Fr[2];
#pragma unroll
                            for (int db = 0; db < 2; ++db) {
                                const float f0 = my_w[((ra * 3 + rb + db) * 3 + q0 + dq) * BLOCK];
                                const float f1 = my_w[(((ra + 1) * 3 + rb + db) * 3 + q0 + dq) * BLOCK];
                                Fr[db] = __builtin_fmaf(tA, f1 - f0, f0);
                            }
                            X[dq] = __builtin_fmaf(tB, Fr[1] - Fr[0], Fr[0]);
                        }
                        xe[w] = X[0];
                        xd[w] = X[1] - X[0];
                    }
#pragma unroll
                    for (int j = 0; j < kUwIn; ++j) {
                        if (j < m_in) {
                            const bool second = j >= jc;
                            const float rj = (j & 1) ? rp[j >> 1].y : rp[j >> 1].x, tj = (j & 1) ? tp[j >> 1].y : tp[j >> 1].x;
                            const float tot = (g + rj) + __builtin_fmaf(tj, second ? xd[1] : xd[0], second ? xe[1] : xe[0]);
                            if (!found && has && tot == best) { best_j = j; best_o1 = o1; found = true; }
                        }
                    }
                }
                label = o0 + P->m[0] * (best_o1 + P->m[1] * best_j);
            }
        } else {
            // ---- a point outside the usual shape: every backup on its own, from the tables (rare by the host's choice) -------------
            bool first = true;
            for (int o0 = 0; o0 < m_o0; ++o0) {
                const int cA = uw_lane(rec0, kUwA + 2 * o0);
                tw[AX_A] = uw_lanef(rec0, kUwA + 2 * o0 + 1);
                float go0 = gpre;
                if (cl0_present) {
                    const float x = s_ot[N->ot[CL0].lds_off + o0 * N->ot[CL0].c0];
                    go0 = cl0_first ? x : go0 + x;
                }
                for (int o1 = 0; o1 < m_o1; ++o1) {
                    const int cB = uw_lane(rec0, kUwB + 2 * o1);
                    tw[AX_B] = uw_lanef(rec0, kUwB + 2 * o1 + 1);
                    float g = go0;
                    if (cl1_present) {
                        const float x = s_ot[cl1_off + o0 * cl1_c0 + o1 * cl1_c1];
                        g = cl1_first ? x : g + x;
                    }
                    const int ob = pbase + js[AX_A] * cA + js[AX_B] * cB;
                    int prev = -0x7fffffff;
                    float e0 = 0.f, de = 0.f;
                    for (int j = 0; j < m_in; ++j) {
                        int lc = uw_lane(rec1, kUwC - 64 + j) - plane0;
                        if (lc < 0 || lc + 1 >= nplanes) {
                            *P->status = 1;
                            lc = lc < 0 ? 0 : nplanes - 2;
                        }
                        if (lc != prev) {
                            const TJ *Jp = Jn + (int64_t)js[D - 1] * lc;
                            e0 = contract_df<TJ, D - 1, D>(Jp, ob, js, tw);
                            de = contract_df<TJ, D - 1, D>(Jp + js[D - 1], ob, js, tw) - e0;
                            prev = lc;
                        }
                        const float tot = (g + uw_lanef(rec1, kUwR - 64 + j)) + __builtin_fmaf(uw_lanef(rec1, kUwT - 64 + j), de, e0);
                        if (first || tot < best) {
                            best = tot;
                            label = o0 + P->m[0] * (o1 + P->m[1] * j);
                            first = false;
                        }
                    }
                }
            }
        }
        if (valid) {
            const int64_t in_plane = (int64_t)ii + (int64_t)inner * (ia + nA * ib);
            const int64_t inner_sz = P->inner;
            Jout[in_plane + inner_sz * (ic + P->halo_lo)] = (TJ)best;
            if (idx_out) st_idx(idx_out, in_plane + inner_sz * ic, label + P->index_base, P->idx_bytes);
        }
    }
}

}  // namespace hjb
