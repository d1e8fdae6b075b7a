// kernels_colcoop.h - variant 7, cooperative form: the column sweep of kernels_colsweep.h with the corner rows of
// EIGHT neighbouring columns fetched once and shared through LDS.
//
// What bounds the one-wave-per-column kernel (profiles/r02_c4_colsweep_v2_pmc.json): every wave gathers its own 30
// corner rows per step, one step ahead at most (the rows sit in registers), while a row is needed by ~30 columns; the
// L1 hit rate is 65 % because those columns run in other waves at other times, and almost every half step waits for one
// access that went to HBM.  Columns that are neighbours along the WINDOW axis and share the group-axis index need the
// same group-axis cells (pos-att: w+ depends on w and the thrusters only) and overlapping window knots (v moves < 1
// cell), and their axis-1 cell sequence is the same (theta+ = theta + h w does not involve v).  So:
//
//   * a workgroup = W = 8 waves = the columns iw0 .. iw0 + 7 of one (group-axis index, axis-0 chunk of 64 states); it
//     sweeps i1 in lockstep;
//   * per step the workgroup stages, for each of its <= NG distinct group-axis cells, the two group rows x NV = W + 3
//     window knots x XW = 72 axis-0 knots - 110 rows instead of the 240 its waves would gather - with 16-byte loads
//     (4 per thread and step instead of 30), into one of two LDS buffers; the loads for step i1 + 1 are issued before
//     step i1 is computed and written to LDS after it: a full step of latency cover, one barrier per step;
//   * a lane reads the two axis-0 neighbours of a corner row from LDS at its OWN cell (ds_read_b32 at a per-group
//     address + immediate offsets): no DPP, no constraint on the axis-0 cells, 64 states per wave.
//
// Arithmetic, plans and member slots are those of the column sweep (cs_group): bit-identical results.
// Eligibility is decided on the host (hjbdp_setup.hip::colcoop_plan): axis 1's cell must not depend on the window-axis index,
// every workgroup's columns must fit the staged ranges, n0 must be a multiple of the staging load width.
#pragma once
#include "kernels_colsweep.h"

namespace hjb {

constexpr int kCcW = 8;                 // columns (waves) per workgroup
constexpr int kCcNV = kCcW + 3;         // window knots staged per group-axis cell
constexpr int kCcXW = 72;               // axis-0 knots staged per row: 64 states' cells + the upper neighbour, the spread of the
constexpr int kCcXWh = 80;              // columns' shifts and the alignment of the first one to a staging load (float16: 8 knots)
constexpr int kCcNCG = 5;               // group-axis cells per workgroup (= the largest NG this form is built for)
// Per-workgroup words: [0] first staged axis-0 knot  [1] group-axis cells in use
//   [2 + c] byte offset (in J, at axis-1 knot 0) of cell c's lower group row at its first staged window knot
//   [2 + NCG + c] number of staged window knots of cell c that exist (the others repeat the last one; never read)
constexpr int kCcWgWords = 16;
static_assert(2 + 2 * kCcNCG <= kCcWgWords, "workgroup words");

template <typename T, typename TJ, int GAX, int NG, bool FASTCOST, int EPL>
__global__ void __launch_bounds__(kCcW * 64)
k_backup_colcoop(const DParams *__restrict__ P, const DTabled *__restrict__ TB, const DColSweep *__restrict__ CS,
                 const TJ *__restrict__ Jn, TJ *__restrict__ Jout, void *__restrict__ idx_out) {
    static_assert(sizeof(T) == 4, "float32 arithmetic");
    static_assert(NG <= kCcNCG, "staged group-axis cells");
    constexpr int D = 4, NW = kCsNW, MM = kCsMMax, W = kCcW, NV = kCcNV, XW = sizeof(TJ) == 2 ? kCcXWh : kCcXW;
    constexpr int ESZ = (int)sizeof(TJ);
    constexpr int ROWB = XW * ESZ;                          // bytes of a staged row
    constexpr int KB = NV * ROWB;                           // lower -> upper group row of a cell
    constexpr int SB = NG * 2 * KB;                         // bytes of one stage buffer
    constexpr int LB = EPL * ESZ;                           // bytes per staging load: 16, or 4 (n0 not a multiple of 4)
    constexpr int NJ = SB / LB;                             // staging loads per step and workgroup
    constexpr int NQ = (NJ + W * 64 - 1) / (W * 64);        // ... per thread
    static_assert(XW % EPL == 0 && (LB == 16 || LB == 4), "staging load width");
    typedef cs_f4 f4;
    typedef cs_f2 f2;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) char lds_char;
    __shared__ __attribute__((aligned(16))) char s_stage[2 * SB];
    __shared__ f4 s_slots[W][NG * MM * 2];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int n0 = P->n[0], n1 = P->n[1], n2 = P->n[2], n3 = P->n[3];
    const int chunks = (n0 + 63) / 64;
    // ---- which columns (workgroup b serves XCD b % 8, see DColSweep::xcd_ig) ---------------------------------
    const int nwax = GAX == 3 ? n2 : n3;                    // columns along the window axis
    const int nblk = (nwax + W - 1) / W;
    int ig, chunk, blk;
    {
        const unsigned xcd = blockIdx.x & 7u, item = blockIdx.x >> 3;
        const unsigned cnt = (unsigned)CS->xcd_cnt[xcd];
        if (item >= cnt * (unsigned)chunks * (unsigned)nblk) return;          // uniform over the workgroup
        ig = as_const<int32_t>(CS->xcd_ig)[xcd * (unsigned)CS->xcd_stride + item % cnt];
        const unsigned r = item / cnt;
        chunk = (int)(r % (unsigned)chunks);
        blk = (int)(r / (unsigned)chunks);
    }
    int iw = blk * W + wave;
    const bool live = iw < nwax;                            // a wave past the end repeats the last column, stores nothing
    if (!live) iw = nwax - 1;
    const int i2 = GAX == 3 ? iw : ig, i3 = GAX == 3 ? ig : iw;
    int i0 = chunk * 64 + lane;
    const bool valid = live && i0 < n0;
    if (i0 >= n0) i0 = n0 - 1;
    cptr<int32_t> wg = as_const<int32_t>(CS->wg) + (size_t)((ig * chunks + chunk) * nblk + blk) * kCcWgWords;
    const int xlo = wg[0];
    // ---- the plan of this column: header words in scalar registers, member slots parked in LDS ------------
    cptr<int32_t> pl = as_const<int32_t>(CS->plan) + (size_t)(i2 + n2 * i3) * kCsPlanWords;
    {
        gptr<f4> src = as_global<f4>(CS->plan + (size_t)(i2 + n2 * i3) * kCsPlanWords + kCsPI);
#pragma unroll
        for (int e = lane; e < NG * MM * 2; e += 64) s_slots[wave][e] = src[e];
    }
    const int ng = pl[0] >> 8;
    if ((pl[0] & 1) && lane == 0) *P->status = 1;
    uint32_t lrow[NG];
    int used[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        lrow[g] = (uint32_t)pl[1 + 2 * kCsGMax + g];
        used[g] = pl[1 + kCsGMax + g];
    }
    // ---- axis 0: the thread's own (cell, t) for the whole column; its byte offset inside a staged row -------
    uint32_t lanebase;
    T t0;
    {
        const DTabled::Axis &A0 = TB->ax[0];
        const int off = A0.sstride[0] * i0 + A0.sstride[2] * i2 + A0.sstride[3] * i3;
        const int c0 = as_global<TabEntry<T>>(A0.tab)[off].cell;
        t0 = as_global<TabEntry<T>>(A0.tab)[off].t;
        lanebase = (uint32_t)(c0 - xlo) * (uint32_t)ESZ;
    }
    // ---- this thread's staging loads: which (cell, group row, window knot, axis-0 segment), fixed for the sweep ----
    const uint32_t g_bytes = CS->g_bytes, w_bytes = CS->w_bytes, s1_bytes = CS->s1_bytes;
    uint32_t soff[NQ], sdst[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int j = q * (W * 64) + (int)threadIdx.x;
        if (j > NJ - 1) j = NJ - 1;                          // the tail repeats the last load (same data, same place)
        constexpr int SEG = XW / EPL;
        const int r = j / SEG, seg = j % SEG;
        int c = r / (2 * NV);
        const int k = (r / NV) % 2, vk = r % NV;
        if (c > wg[1] - 1) c = wg[1] - 1;                    // cells not in use repeat the last one (never read)
        int nv = 0;
        uint32_t coff = 0;
#pragma unroll
        for (int cc = 0; cc < NG; ++cc)
            if (cc == c) { coff = (uint32_t)wg[2 + cc]; nv = wg[2 + kCcNCG + cc]; }
        const int vkc = vk < nv ? vk : nv - 1;
        int x = xlo + seg * EPL;
        if (x > n0 - EPL) x = n0 - EPL;                      // past the row's end (n0 % EPL == 0): never read
        soff[q] = coff + (uint32_t)k * g_bytes + (uint32_t)vkc * w_bytes + (uint32_t)x * (uint32_t)ESZ;
        sdst[q] = (uint32_t)j * (uint32_t)LB;
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
    const uint32_t out_col = (uint32_t)i0 + (uint32_t)P->jstride[2] * (uint32_t)i2 + (uint32_t)P->jstride[3] * (uint32_t)(i3 + P->halo_lo);
    const uint32_t idx_col = (uint32_t)i0 + (uint32_t)n0 * (uint32_t)n1 * ((uint32_t)i2 + (uint32_t)n2 * (uint32_t)i3);
    const uint32_t js1 = (uint32_t)P->jstride[1];
    const int index_base = P->index_base, idx_bytes = P->idx_bytes;
    lds_char *stage = (lds_char *)&s_stage[0];
    typedef __attribute__((address_space(3))) const f4 lds_f4;
    lds_f4 *slots = (lds_f4 *)&s_slots[wave][0];
    asm volatile("" : "+v"(slots));          // one address register for the column, not one re-made per read

    // staging: request the rows of axis-1 knot `knot` / park them in stage buffer `b`
    u4 R4[LB == 16 ? NQ : 1];
    uint32_t R1[LB == 4 ? NQ : 1];
    auto stage_load = [&](int knot) {
        gptr<char> base = as_global<char>(reinterpret_cast<const char *>(Jn) + (size_t)knot * s1_bytes);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (LB == 16) R4[q] = *reinterpret_cast<gptr<u4>>(base + soff[q]);
            else R1[q] = *reinterpret_cast<gptr<uint32_t>>(base + soff[q]);
        }
    };
    auto stage_park = [&](int b) {
        lds_char *dst = stage + b * SB;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (LB == 16) *reinterpret_cast<__attribute__((address_space(3))) u4 *>(dst + sdst[q]) = R4[q];
            else *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(dst + sdst[q]) = R1[q];
        }
    };
    // the two axis-0 neighbours of the corner row (group g, group row k, window knot w) in stage buffer b
    auto knot_lo = [&](uint32_t gaddr, int k, int w) -> T {
        return (T) * reinterpret_cast<__attribute__((address_space(3))) const TJ *>(stage + gaddr + (k * KB + w * ROWB));
    };
    auto knot_hi = [&](uint32_t gaddr, int k, int w) -> T {
        return (T) * reinterpret_cast<__attribute__((address_space(3))) const TJ *>(stage + gaddr + (k * KB + w * ROWB + ESZ));
    };

    f2 A[NG][NW];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int w = 0; w < NW; ++w) A[g][w] = f2{(T)0, (T)0};
    int prev_c1 = -2;
    int c1n = tab1[0].cell;
    T t1n = tab1[0].t;
    stage_load(c1n + 1);                                     // prologue: the rows of step 0
    stage_park(0);
    __syncthreads();
    T best = (T)0, gstep;
    int best_u = 0;
    for (int i1 = 0; i1 < n1; ++i1) {
        const int c1 = c1n;
        const T t1 = t1n;
        const int b = i1 & 1;
        int ngs = ng;
        asm volatile("" : "+s"(ngs));        // group-count tests stay scalar compares of this step
        {   // next step's axis-1 entry: a scalar load in flight during this step
            const int nx = (i1 + 1 < n1 ? i1 + 1 : i1) * a1_s;
            c1n = tab1[nx].cell;
            t1n = tab1[nx].t;
        }
        if (c1 != prev_c1 + 1) {         // (re-)prime: A <- the rows at knot c1 (column start; irregular axis-1 cells).
            stage_load(c1);              // The same for every wave of the workgroup: axis 1 does not see the window axis.
            stage_park(b ^ 1);
            __syncthreads();
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (g < ng) {
                    const uint32_t ga = lanebase + (lrow[g] + (uint32_t)((b ^ 1) * SB));
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const f2 l = {knot_lo(ga, 0, w), knot_lo(ga, 1, w)}, h = {knot_hi(ga, 0, w), knot_hi(ga, 1, w)};
                        const f2 t0p = {t0, t0};
                        A[g][w] = __builtin_elementwise_fma(t0p, h - l, l);
                    }
                }
            __syncthreads();
        }
        prev_c1 = c1;
        stage_load(c1n + 1);                                  // the rows of the next step: a whole step to land
        if (i1 > 0 && valid) {                                // the previous step's results
            stj<T, TJ>(Jout, (int64_t)(out_col + js1 * (uint32_t)(i1 - 1)), best);
            if (idx_out) st_idx(idx_out, idx_col + (uint32_t)n0 * (uint32_t)(i1 - 1), best_u + index_base, idx_bytes);
        }
        // ---- this state's cost without the control terms -----------------------------------------------
        gstep = gcol;
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
        best = __builtin_inff();                 // (inf, control 0): what an all-infinite column of totals yields as well
        best_u = 0;
        // the corner values of group g + 1 are requested from LDS before group g is computed
        TJ cv[2][2][NW][2];                                   // [buffer][group row][window knot][lower, upper neighbour]
        auto read_group = [&](int g, int q) {
            const uint32_t ga = lanebase + (lrow[g] + (uint32_t)(b * SB));
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    cv[q][k][w][0] = *reinterpret_cast<__attribute__((address_space(3))) const TJ *>(stage + ga + (k * KB + w * ROWB));
                    cv[q][k][w][1] = *reinterpret_cast<__attribute__((address_space(3))) const TJ *>(stage + ga + (k * KB + w * ROWB + ESZ));
                }
        };
        read_group(0, 0);                                     // every column has a group 0
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g < ngs) {
                if (g + 1 < NG) read_group(g + 1, (g + 1) & 1);     // unconditionally (a padded group repeats group 0's rows): the
                                                                    // waits for group g then count exactly the younger reads
                // a lane's two neighbours arrive as a register pair (one ds_read2_b32); the (lower, upper group row) pair
                // the packed arithmetic works on is made by the two plain lerps' destinations, not by moves
                auto row0 = [&](int w) {
                    const T l0 = (T)cv[g & 1][0][w][0], l1 = (T)cv[g & 1][1][w][0];
                    T a0 = fma_t<T>(t0, (T)((T)cv[g & 1][0][w][1] - l0), l0);
                    asm volatile("" : "+v"(a0));
                    const T a1 = fma_t<T>(t0, (T)((T)cv[g & 1][1][w][1] - l1), l1);
                    return f2{a0, a1};
                };
                cs_group<T, GAX, FASTCOST>(used[g], g, row0, A[g], A[g], t1, slots, gstep, ncu, npre, best, best_u);
            }
        }
        stage_park(b ^ 1);
        __syncthreads();
    }
    if (valid) {
        stj<T, TJ>(Jout, (int64_t)(out_col + js1 * (uint32_t)(n1 - 1)), best);
        if (idx_out) st_idx(idx_out, idx_col + (uint32_t)n0 * (uint32_t)(n1 - 1), best_u + index_base, idx_bytes);
    }
}

}  // namespace hjb
