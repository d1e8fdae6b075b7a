// kernels_colsweep.h - variant 7: the column-sweep stage kernel for the pos-att shape (C4 / C5).
//
// Shape (checked on the host, hjbdp.hip::ensure_colsweep): D = 4, one control dim, and
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
// and A does not depend on the state's index along axis 1.  So one WAVE = 64 consecutive axis-0 states of one
// (i2, i3) pair, and it SWEEPS THE COLUMN i1 = 0..n1-1: the A row at knot c1+1 of one step is the A row at knot c1
// of the next (axis-1 cells advance by one per state when the displacement is sub-cell; anything else re-primes,
// a wave-uniform branch).  Per step a lane loads each needed corner row once (2 loads), not once per control.
//
// Which (k2, k3) rows a state needs is stage-invariant and identical for the whole column, so the host builds a PLAN
// per (i2, i3): the controls are put into GROUPS sharing the cell of the "group axis" GAX (pos-att: w, 5 distinct
// cells among the 9 thruster combinations, Solver_pos_att.m:886-904) whose cells along the other, "window" axis span
// at most NW knots (v moves < 1 cell).  A group = 2 x NW corner rows; its members differ only in weights, cost and
// which window pair they use.  The kernel is a template over the number of groups NG (the maximum over the plans;
// plans with fewer are padded with member-less groups) and straight-line over groups and rows: all 4 NG NW loads of a
// step are issued back to back, the rolling A values sit in registers with static indices, and everything per control
// is wave-uniform data of the plan, parked in LDS for the column and read back by broadcast.  Groups are visited in
// plan order, not control order: a member that is visited after a higher-numbered control carries a tie flag and
// compares (value, index), so the first-index-wins rule of MATLAB's min holds exactly.  Same canonical arithmetic:
// bit-identical to every other variant.
#pragma once
#include "hjbdp_dev.h"
#include "kernels_generic.h"
#include "kernels_tabled.h"
#include "kernels_rowwise.h"

namespace hjb {

constexpr int kCsGMax = 6;      // groups per (i2, i3)
constexpr int kCsMMax = 3;      // members per group
constexpr int kCsNW = 3;        // window knots per group (cells span <= NW - 1)
constexpr int kCsUMax = 16;     // controls
constexpr int kCsMaxCu = kLeanMaxCu;
constexpr int kCsAhead = 2;     // corner-row loads are issued this many groups ahead of their use
// The plan of one (i2, i3), 32-bit words:
//   [0] halo violation flag   [1 + g] byte offset of group g's first corner row   [1 + GMAX + g] nw | nmem << 8
//   [kCsPI + 8 s ...] member slot s = g * MMAX + ms: t_window, t_group, cu[0], u | off << 8 | tie << 16, cu[1..3], 0
constexpr int kCsPI = 16;
constexpr int kCsSlots = kCsGMax * kCsMMax;
constexpr int kCsPlanWords = kCsPI + 8 * kCsSlots;
static_assert(1 + 2 * kCsGMax <= kCsPI, "plan header");

struct DColSweep {
    const int32_t *plan;
    int32_t gax;            // group axis: 2 or 3 (the other one is the window axis)
    int32_t ng;             // groups per plan (maximum over the plans)
    int32_t npre_col;       // leading state-only cost terms that do not depend on state dim 1: summed once per column
    int32_t step_uniform;   // the remaining state-only cost terms do not depend on state dim 0 (wave-uniform per step)
    int32_t ncu;            // control-only cost terms
    uint32_t g_bytes;       // J byte stride of the group axis
    uint32_t w_bytes;       // J byte stride of the window axis
    uint32_t s1_bytes;      // J byte stride of axis 1
    int32_t tile2, tile3;   // traversal tile over (i2, i3): consecutive waves share corner rows
};

template <typename T, typename TJ, int GAX, int NG, bool FASTCOST>
__global__ void __launch_bounds__(256)
k_backup_colsweep(const DParams *__restrict__ P, const DTabled *__restrict__ TB, const DColSweep *__restrict__ CS,
                  const TJ *__restrict__ Jn, TJ *__restrict__ Jout, int32_t *__restrict__ idx_out) {
    static_assert(sizeof(T) == 4, "float32 arithmetic");
    constexpr int D = 4, NW = kCsNW, MM = kCsMMax;
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ f4 s_slots[4][kCsSlots * 2];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int n0 = P->n[0], n1 = P->n[1], n2 = P->n[2], n3 = P->n[3];
    const int chunks = (n0 + 63) >> 6;
    // ---- which column: tiles of (i2, i3), i2 fastest inside a tile --------------------------------------
    const int t2 = CS->tile2, t3 = CS->tile3;
    const int nt2 = (n2 + t2 - 1) / t2;
    int i2, i3, chunk;
    {
        const unsigned item = blockIdx.x * 4u + (unsigned)wave;
        chunk = (int)(item % (unsigned)chunks);
        const unsigned r = item / (unsigned)chunks;
        const unsigned tile = r / (unsigned)(t2 * t3), within = r % (unsigned)(t2 * t3);
        i2 = (int)(tile % (unsigned)nt2) * t2 + (int)(within % (unsigned)t2);
        i3 = (int)(tile / (unsigned)nt2) * t3 + (int)(within / (unsigned)t2);
    }
    if (i2 >= n2 || i3 >= n3) return;                       // ragged tiles (uniform over the wave)
    int i0 = chunk * 64 + lane;
    const bool valid = i0 < n0;
    if (!valid) i0 = n0 - 1;                                // duplicate work, no store
    // ---- the plan of this column: header words in scalar registers, member slots parked in LDS ------------
    cptr<int32_t> pl = as_const<int32_t>(CS->plan) + (size_t)(i2 + n2 * i3) * kCsPlanWords;
    {
        gptr<f4> src = as_global<f4>(CS->plan + (size_t)(i2 + n2 * i3) * kCsPlanWords + kCsPI);
        if (lane < NG * MM * 2) s_slots[wave][lane] = src[lane];
    }
    if (pl[0] && lane == 0) *P->status = 1;
    const uint32_t g_bytes = CS->g_bytes, w_bytes = CS->w_bytes, s1_bytes = CS->s1_bytes;
    uint32_t roff[NG][NW];
    int nmem[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const uint32_t ro = (uint32_t)pl[1 + g];
        const int info = pl[1 + kCsGMax + g];
        const int nw = info & 0xff;
        nmem[g] = (info >> 8) & 0xff;
#pragma unroll
        for (int w = 0; w < NW; ++w) roff[g][w] = ro + (uint32_t)(w < nw ? w : nw - 1) * w_bytes;   // a short window re-reads its last row
    }
    // ---- axis 0: the thread's own (cell, t) for the whole column ------------------------------------------
    uint32_t voff0;
    T t0;
    {
        const DTabled::Axis &A0 = TB->ax[0];
        const int off = A0.sstride[0] * i0 + A0.sstride[2] * i2 + A0.sstride[3] * i3;
        voff0 = (uint32_t)as_global<TabEntry<T>>(A0.tab)[off].cell * (uint32_t)sizeof(TJ);
        t0 = as_global<TabEntry<T>>(A0.tab)[off].t;
    }
    const DTabled::Axis &A1 = TB->ax[1];
    const int a1_base = A1.sstride[2] * i2 + A1.sstride[3] * i3, a1_s = A1.sstride[1];
    cptr<TabEntry<T>> tab1 = as_const<TabEntry<T>>(A1.tab) + a1_base;
    const int ncu = CS->ncu, npre_col = CS->npre_col, npre = P->n_cost_prefix;
    const bool step_uniform = CS->step_uniform != 0;
    // ---- cost: leading state-only terms that do not change along the column --------------------------------
    int si[D] = {i0, 0, i2, i3 + P->slab_begin};
    const int cjz[HJB_MAX_C] = {0, 0, 0};
    T gcol = (T)0;
    for (int k = 0; k < npre_col; ++k) {
        const T x = term_value<T, D>(P->cost[k], si, cjz);
        gcol = (k == 0) ? x : (T)(gcol + x);
    }
    // Four scalar bases - (lower, upper) group row x (lower, upper) axis-0 neighbour - and ONE 32-bit per-lane byte
    // offset per (group, window knot): every gather is `global_load v, v_off, s[base]`, no 64-bit vector arithmetic.
    // The bases are opaque to the compiler: the two loads of a corner pair must stay two instructions (merged into one
    // unaligned 8-byte load they are slower, measured on the row kernel).
    gptr<char> Jb00 = as_global<char>(Jn);
    gptr<char> Jb01 = as_global<char>(reinterpret_cast<const char *>(Jn) + sizeof(TJ));
    gptr<char> Jb10 = as_global<char>(reinterpret_cast<const char *>(Jn) + g_bytes);
    gptr<char> Jb11 = as_global<char>(reinterpret_cast<const char *>(Jn) + g_bytes + sizeof(TJ));
    asm volatile("" : "+s"(Jb01));
    asm volatile("" : "+s"(Jb10));
    asm volatile("" : "+s"(Jb11));
    const uint32_t out_col = (uint32_t)i0 + (uint32_t)P->jstride[2] * (uint32_t)i2 + (uint32_t)P->jstride[3] * (uint32_t)(i3 + P->halo_lo);
    const uint32_t idx_col = (uint32_t)i0 + (uint32_t)n0 * (uint32_t)n1 * ((uint32_t)i2 + (uint32_t)n2 * (uint32_t)i3);
    const uint32_t js1 = (uint32_t)P->jstride[1];
    const int index_base = P->index_base;
    __builtin_amdgcn_wave_barrier();

    T A[NG][2][NW];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int w = 0; w < NW; ++w) A[g][k][w] = (T)0;
    int prev_c1 = -2;
    int c1n = tab1[0].cell;
    T t1n = tab1[0].t;
    for (int i1 = 0; i1 < n1; ++i1) {
        const int c1 = c1n;
        const T t1 = t1n;
        {   // next step's axis-1 entry: a scalar load in flight during this step
            const int nx = (i1 + 1 < n1 ? i1 + 1 : i1) * a1_s;
            c1n = tab1[nx].cell;
            t1n = tab1[nx].t;
        }
        if (c1 != prev_c1 + 1) {         // (re-)prime: A <- the row at knot c1 (column start; irregular axis-1 cells)
            const uint32_t vrow = voff0 + (uint32_t)c1 * s1_bytes;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const uint32_t o = roff[g][w] + vrow;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const T lo = (T) * reinterpret_cast<gptr<TJ>>((k ? Jb10 : Jb00) + o);
                        const T hi = (T) * reinterpret_cast<gptr<TJ>>((k ? Jb11 : Jb01) + o);
                        A[g][k][w] = fma_t<T>(t0, (T)(hi - lo), lo);
                    }
                }
        }
        prev_c1 = c1;
        // ---- the corner rows at knot c1 + 1: every load of the step, back to back ----------------------------
        const uint32_t vrow = voff0 + (uint32_t)(c1 + 1) * s1_bytes;
        // loads run kCsAhead groups ahead of the arithmetic (software pipeline over the straight-line group sequence)
        T lo[NG][2][NW], hi[NG][2][NW];
        auto load_group = [&](int g) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t o = roff[g][w] + vrow;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    lo[g][k][w] = (T) * reinterpret_cast<gptr<TJ>>((k ? Jb10 : Jb00) + o);
                    hi[g][k][w] = (T) * reinterpret_cast<gptr<TJ>>((k ? Jb11 : Jb01) + o);
                }
            }
        };
#pragma unroll
        for (int g = 0; g < kCsAhead && g < NG; ++g) load_group(g);
        // ---- this state's cost without the control terms -----------------------------------------------
        T gstep = gcol;
        if (npre > npre_col) {
            si[1] = i1;
            if (step_uniform) {
                for (int k = npre_col; k < npre; ++k) {
                    const DTerm &tm = P->cost[k];
                    const int off = tm.stride[1] * i1 + tm.stride[2] * i2 + tm.stride[3] * si[3];
                    const T x = as_const<T>(tm.data)[off];
                    gstep = (k == 0) ? x : (T)(gstep + x);
                }
            } else {
                for (int k = npre_col; k < npre; ++k) {
                    const T x = term_value<T, D>(P->cost[k], si, cjz);
                    gstep = (k == 0) ? x : (T)(gstep + x);
                }
            }
        }
        T best = (T)0;
        int best_u = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + kCsAhead < NG) {
                load_group(g + kCsAhead);
                __builtin_amdgcn_sched_barrier(0);
            }
            f4 m0[MM], m1[MM];
#pragma unroll
            for (int ms = 0; ms < MM; ++ms) {
                m0[ms] = s_slots[wave][(g * MM + ms) * 2];
                if (!FASTCOST && ncu > 1) m1[ms] = s_slots[wave][(g * MM + ms) * 2 + 1];
            }
            T Bv[2][NW];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const T an = fma_t<T>(t0, (T)(hi[g][k][w] - lo[g][k][w]), lo[g][k][w]);
                    Bv[k][w] = fma_t<T>(t1, (T)(an - A[g][k][w]), A[g][k][w]);
                    A[g][k][w] = an;
                }
#pragma unroll
            for (int ms = 0; ms < MM; ++ms) {
                if (ms < nmem[g]) {
                    const int minfo = __builtin_amdgcn_readfirstlane(__float_as_int(m0[ms].w));
                    const int u = minfo & 0xff, off = (minfo >> 8) & 0xff, tie = minfo >> 16;
                    const T tw = m0[ms].x, tg = m0[ms].y;
                    // the member's 2 x 2 corners of B: window knots off, off + 1 of both group rows
                    T b00, b01, b10, b11;                        // [k][window lo/hi]
                    if (off == 0) { b00 = Bv[0][0]; b01 = Bv[0][1]; b10 = Bv[1][0]; b11 = Bv[1][1]; }
                    else          { b00 = Bv[0][1]; b01 = Bv[0][2]; b10 = Bv[1][1]; b11 = Bv[1][2]; }
                    T interp;
                    if (GAX == 3) {          // window = axis 2 (lerped first), group = axis 3
                        const T v0 = fma_t<T>(tw, (T)(b01 - b00), b00);
                        const T v1 = fma_t<T>(tw, (T)(b11 - b10), b10);
                        interp = fma_t<T>(tg, (T)(v1 - v0), v0);
                    } else {                 // group = axis 2 (lerped first), window = axis 3
                        const T v0 = fma_t<T>(tg, (T)(b10 - b00), b00);
                        const T v1 = fma_t<T>(tg, (T)(b11 - b01), b01);
                        interp = fma_t<T>(tw, (T)(v1 - v0), v0);
                    }
                    T gg;
                    if (FASTCOST) {                              // the usual shape: state terms + ONE control term
                        gg = (T)(gstep + m0[ms].z);
                    } else {
                        gg = gstep;
                        for (int k = 0; k < ncu; ++k) {
                            const T x = k == 0 ? m0[ms].z : (k == 1 ? m1[ms].x : (k == 2 ? m1[ms].y : m1[ms].z));
                            gg = (npre == 0 && k == 0) ? x : (T)(gg + x);
                        }
                    }
                    const T tot = (T)(gg + interp);
                    bool take;
                    if (g == 0 && ms == 0) take = true;
                    else if (tie) take = tot < best || (tot == best && u < best_u);
                    else take = tot < best;
                    if (take) { best = tot; best_u = u; }
                }
            }
        }
        if (valid) {
            stj<T, TJ>(Jout, (int64_t)(out_col + js1 * (uint32_t)i1), best);
            if (idx_out) idx_out[idx_col + (uint32_t)n0 * (uint32_t)i1] = best_u + index_base;
        }
    }
}

}  // namespace hjb
