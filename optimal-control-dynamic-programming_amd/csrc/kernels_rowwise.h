// kernels_rowwise.h - variant 6: variant 5 (per-axis (cell, t) tables) with one WAVEFRONT PER GRID ROW.
//
// Applies when no axis other than axis 0 depends on state dim 0 - pos-att (Solver_pos_att.m:299-328: v+ over
// (v,u), theta+ over (theta,w), w+ over (w,u)), the 2-D channels of Solver_position / Solver_attitude.  A wave
// then holds <= 64 consecutive axis-0 states of ONE row, so for a given control the cells and weights of axes
// 1..D-1 are the same for all its lanes: they are read with scalar loads, the 2^(D-1) corner-row base addresses
// are scalar arithmetic, and a lane's 2^D gathers are `row base (SGPR pair) + 4 * cell0 (VGPR) [+ 4]` - no
// per-lane 64-bit address arithmetic, no per-lane table lookups, no per-lane index decomposition.  Measured on C4
// (120^4 x 9): variant 5 spends 144 VALU instructions per backup, most of them on addresses (profiles/
// r01_c4_pmc.json); this form needs ~50 and runs C4 10 % faster - the gather path itself (16 loads per backup) is
// what remains.  Staging each corner row in LDS with one coalesced load per row was tried and was 2.3x SLOWER
// (load -> LDS write -> LDS read is one long dependent chain per control).  Same canonical arithmetic and lerp order: bit-identical to every other variant.
// Any D <= 6, C <= 3, float32/float64/float16-storage, slabs.
#pragma once
#include <type_traits>
#include "hjbdp_dev.h"
#include "kernels_generic.h"
#include "kernels_tabled.h"

namespace hjb {

// read-only tables through the constant address space: a wave-uniform index becomes a scalar load
template <typename T> using cptr = const __attribute__((address_space(4))) T *;
template <typename T> __device__ __forceinline__ cptr<T> as_const(const void *p) { return (cptr<T>)p; }

__device__ __forceinline__ float uniform_value(float x) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
__device__ __forceinline__ double uniform_value(double x) {
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <typename T, typename TJ, int D>
__global__ void __launch_bounds__(256)
k_backup_rowwise(const DParams *__restrict__ P, const DTabled *__restrict__ TB, const TJ *__restrict__ Jn,
                 TJ *__restrict__ Jout, void *__restrict__ idx_out) {
    static_assert(D >= 2, "one wave per row needs a second axis");
    constexpr int DR = D - 1;                         // axes 1..D-1: wave-uniform
    constexpr int NR = 1 << (D - 1);                  // corner rows
    const int C = P->C;
    const int nU = (int)P->nU;
    const int plane0 = P->plane0, nplanes = P->nplanes;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int n0 = P->n[0];
    const int chunks = (n0 + 63) >> 6;
    int64_t rows = 1;
#pragma unroll
    for (int a = 1; a < D; ++a) rows *= P->n[a];
    const int64_t items = rows * chunks;
    const int64_t wstride = (int64_t)gridDim.x * 4;
    const int m1 = P->m[1], m2 = P->m[2];
    int64_t js[D];
#pragma unroll
    for (int a = 0; a < D; ++a) js[a] = P->jstride[a];
    const bool hasc0 = TB->ax[0].has_ctrl != 0;

    for (int64_t item = (int64_t)xcd_share(blockIdx.x, gridDim.x) * 4 + wave; item < items; item += wstride) {      // (XCD-aware: kernels_tabled.h)
        // ---- the row (uniform) and this lane's axis-0 index -------------------------------------
        int sl[D], si[D];
        {
            int64_t r = item;
            const int chunk = (int)(r % chunks);
            r /= chunks;
            sl[0] = chunk * 64 + lane;
#pragma unroll
            for (int a = 1; a < D; ++a) {
                sl[a] = __builtin_amdgcn_readfirstlane((int)(r % P->n[a]));
                r /= P->n[a];
            }
        }
        const bool valid = sl[0] < n0;
        if (!valid) sl[0] = n0 - 1;                   // harmless duplicate work, store skipped
#pragma unroll
        for (int a = 0; a < D; ++a) si[a] = sl[a];
        si[D - 1] += P->slab_begin;
        // table offsets: axis 0 per lane, axes >= 1 uniform
        int aoff0 = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) aoff0 += TB->ax[0].sstride[d] * sl[d];
        int aoffr[DR];
#pragma unroll
        for (int a = 1; a < D; ++a) {
            int off = 0;
#pragma unroll
            for (int d = 1; d < D; ++d) off += TB->ax[a].sstride[d] * sl[d];
            aoffr[a - 1] = off;
        }
        int cell0 = 0;
        T t0 = (T)0;
        if (!hasc0) {
            cell0 = as_global<TabEntry<T>>(TB->ax[0].tab)[aoff0].cell;
            t0 = as_global<TabEntry<T>>(TB->ax[0].tab)[aoff0].t;
        }
        int cjz[HJB_MAX_C] = {0, 0, 0};
        T gpre = (T)0;
        for (int k = 0; k < P->n_cost_prefix; ++k) {
            const T x = term_value<T, D>(P->cost[k], si, cjz);
            gpre = (k == 0) ? x : (T)(gpre + x);
        }
        T best = (T)0;
        int best_u = 0;
        int cj[HJB_MAX_C] = {0, 0, 0};
        for (int u = 0; u < nU; ++u) {
            // ---- axes 1..D-1: scalar (cell, t), corner-row bases --------------------------------
            T tr[DR];
            int64_t rb0 = 0;                          // base of corner row 0 (all lower cells)
            bool bad = false;
#pragma unroll
            for (int a = 1; a < D; ++a) {
                const DTabled::Axis &A = TB->ax[a];
                const int off = aoffr[a - 1] + A.cstride[0] * cj[0] + A.cstride[1] * cj[1] + A.cstride[2] * cj[2];
                // the entry is the same for every lane: pin it to scalar registers so that everything derived
                // from it (row bases, the weights' operand) is scalar arithmetic
                int cl = __builtin_amdgcn_readfirstlane(as_const<TabEntry<T>>(A.tab)[off].cell);
                tr[a - 1] = uniform_value(as_const<TabEntry<T>>(A.tab)[off].t);
                if (a == D - 1) {
                    cl -= plane0;
                    if (cl < 0 || cl + 1 >= nplanes) { bad = true; cl = cl < 0 ? 0 : nplanes - 2; }
                }
                rb0 += js[a] * cl;
            }
            if (bad) *P->status = 1;
            if (hasc0) {
                const DTabled::Axis &A = TB->ax[0];
                const int off = aoff0 + A.cstride[0] * cj[0] + A.cstride[1] * cj[1] + A.cstride[2] * cj[2];
                cell0 = as_global<TabEntry<T>>(A.tab)[off].cell;
                t0 = as_global<TabEntry<T>>(A.tab)[off].t;
            }
            const uint32_t c0 = (uint32_t)cell0;
            // ---- gathers: uniform row base + this lane's axis-0 cell -----------------------------
            T v[1 << D];
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                int64_t rb = rb0;
#pragma unroll
                for (int a = 1; a < D; ++a)
                    if (c & (1 << (a - 1))) rb += js[a];
                const TJ *rowp = Jn + rb;
                v[2 * c] = (T)rowp[c0];
                v[2 * c + 1] = (T)rowp[c0 + 1u];
            }
            // ---- lerps, canonical order: axis 0 (per-lane weight), then axes 1.. (scalar weights) ----
#pragma unroll
            for (int j = 0; j < NR; ++j) v[j] = fma_t<T>(t0, (T)(v[2 * j + 1] - v[2 * j]), v[2 * j]);
#pragma unroll
            for (int a = 1; a < D; ++a) {
#pragma unroll
                for (int j = 0; j < (1 << (D - 1 - a)); ++j)
                    v[j] = fma_t<T>(tr[a - 1], (T)(v[2 * j + 1] - v[2 * j]), v[2 * j]);
            }
            T g = gpre;
            for (int k = P->n_cost_prefix; k < P->n_cost; ++k) {
                const T x = term_value<T, D>(P->cost[k], si, cj);
                g = (k == 0) ? x : (T)(g + x);
            }
            const T tot = (T)(g + v[0]);
            if (u == 0 || tot < best) {
                best = tot;
                best_u = u;
            }
            if (C == 1) {
                ++cj[0];
            } else if (C == 2) {
                if (++cj[1] == m1) { cj[1] = 0; ++cj[0]; }
            } else {
                if (++cj[2] == m2) {
                    cj[2] = 0;
                    if (++cj[1] == m1) { cj[1] = 0; ++cj[0]; }
                }
            }
        }
        if (valid) {
            int64_t label;
            if (C == 1) {
                label = best_u;
            } else if (C == 2) {
                const int j1 = best_u % m1, j0 = best_u / m1;
                label = j0 + (int64_t)P->m[0] * j1;
            } else {
                const int j2 = best_u % m2;
                const int rr = best_u / m2;
                const int j1 = rr % m1, j0 = rr / m1;
                label = j0 + (int64_t)P->m[0] * (j1 + (int64_t)m1 * j2);
            }
            int64_t ls = 0, mul = 1, lj = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                ls += mul * sl[a];
                lj += js[a] * (a == D - 1 ? sl[a] + P->halo_lo : sl[a]);
                mul *= P->n[a];
            }
            stj<T, TJ>(Jout, lj, best);
            if (idx_out) st_idx(idx_out, ls, (int32_t)(label + P->index_base), P->idx_bytes);
        }
    }
}

}  // namespace hjb

namespace hjb {

// ---- lean form ------------------------------------------------------------------------------------------
// Ablation on C4 (removing the gathers, the cost terms and the table lookups in turn changed nothing) showed what
// k_backup_rowwise spends its time on: its own ~270-instruction control loop.  The scalar unit and the vector unit
// of a CU each retire one wave-instruction per cycle, and "scalarising" had only moved 64-bit address arithmetic
// (150 SALU per control) from one to the other.  This form removes the instructions instead:
//   * everything that depends on the CONTROL but not on the lane - row base offset, weights of axes 1..D-1,
//     control-only cost terms - is computed once per row item with LANE u WORKING ON CONTROL u (all controls in
//     parallel, a handful of instructions per item instead of per control) and parked in LDS;
//   * element offsets are 32-bit (J has < 2^31 elements), so a gather is `J + 4*(row + delta_c + cell0)` with the
//     2^(D-1) corner deltas in scalar registers: one v_add per corner row, no 64-bit arithmetic;
//   * the control loop reads its per-control record from LDS (broadcast), gathers, lerps, compares: ~45 VALU.
// Requirements on top of variant 6 (checked on the host): axis 0 independent of the control, nU <= 64, fewer than
// 2^31 J elements, every cost term that involves a control involves controls ONLY (at most kLeanMaxCu of them).
constexpr int kLeanMaxCu = 4;

template <typename T, typename TJ, int D>
__global__ void __launch_bounds__(256)
k_backup_rowlean(const DParams *__restrict__ P, const DTabled *__restrict__ TB, const TJ *__restrict__ Jn,
                 TJ *__restrict__ Jout, void *__restrict__ idx_out) {
    static_assert(D >= 2, "one wave per row needs a second axis");
    constexpr int DR = D - 1;
    constexpr int NR = 1 << (D - 1);
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int C = P->C;
    const int nU = (int)P->nU;
    const int plane0 = P->plane0, nplanes = P->nplanes;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ncu = P->n_cost - P->n_cost_prefix;                   // control-only cost terms
    const bool has_prefix = P->n_cost_prefix > 0;
    // LDS, per wave: rb[nU] (uint32) | tw[nU][DR] (T) | cu[nU][ncu] (T) ; after the four waves: cj[nU][3] (block)
    const size_t rb_bytes = ((size_t)nU * 4 + 15) & ~(size_t)15;
    const size_t per_wave = rb_bytes + (((size_t)nU * (DR + kLeanMaxCu) * sizeof(T) + 15) & ~(size_t)15);
    unsigned char *wbase = smem_raw + (size_t)wave * per_wave;
    uint32_t *s_rb = reinterpret_cast<uint32_t *>(wbase);
    T *s_tw = reinterpret_cast<T *>(wbase + rb_bytes);
    T *s_cu = s_tw + (size_t)nU * DR;
    int *s_cj = reinterpret_cast<int *>(smem_raw + 4 * per_wave);
    // control index -> per-dim indices (control dim 0 slowest), once per workgroup
    for (int u = threadIdx.x; u < nU; u += blockDim.x) {
        int j0 = u, j1 = 0, j2 = 0;
        if (C == 2) { j1 = u % P->m[1]; j0 = u / P->m[1]; }
        if (C == 3) { j2 = u % P->m[2]; const int r = u / P->m[2]; j1 = r % P->m[1]; j0 = r / P->m[1]; }
        s_cj[3 * u] = j0; s_cj[3 * u + 1] = j1; s_cj[3 * u + 2] = j2;
    }
    __syncthreads();
    const uint32_t n0 = (uint32_t)P->n[0];
    const uint32_t chunks = (n0 + 63u) >> 6;
    uint32_t rows = 1;
#pragma unroll
    for (int a = 1; a < D; ++a) rows *= (uint32_t)P->n[a];
    const uint32_t items = rows * chunks;
    const uint32_t wstride = gridDim.x * 4u;
    uint32_t js[D];
#pragma unroll
    for (int a = 0; a < D; ++a) js[a] = (uint32_t)P->jstride[a];
    uint32_t delta[NR];                                             // corner-row offsets (uniform)
#pragma unroll
    for (int c = 0; c < NR; ++c) {
        uint32_t d = 0;
#pragma unroll
        for (int a = 1; a < D; ++a)
            if (c & (1 << (a - 1))) d += js[a];
        delta[c] = d;
    }
    const int my_cj[3] = {lane < nU ? s_cj[3 * lane] : 0, lane < nU ? s_cj[3 * lane + 1] : 0, lane < nU ? s_cj[3 * lane + 2] : 0};

    for (uint32_t item = xcd_share(blockIdx.x, gridDim.x) * 4u + wave; item < items; item += wstride) {      // (XCD-aware: kernels_tabled.h)
        int sl[D], si[D];
        {
            uint32_t r = item;
            const uint32_t chunk = r % chunks;
            r /= chunks;
            sl[0] = (int)(chunk * 64u + lane);
#pragma unroll
            for (int a = 1; a < D; ++a) {
                const uint32_t na = (uint32_t)P->n[a];
                sl[a] = (int)(r % na);
                r /= na;
            }
        }
        const bool valid = (uint32_t)sl[0] < n0;
        if (!valid) sl[0] = (int)n0 - 1;
#pragma unroll
        for (int a = 0; a < D; ++a) si[a] = sl[a];
        si[D - 1] += P->slab_begin;
        int aoff0 = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) aoff0 += TB->ax[0].sstride[d] * sl[d];
        const uint32_t c0 = (uint32_t)as_global<TabEntry<T>>(TB->ax[0].tab)[aoff0].cell;
        const T t0 = as_global<TabEntry<T>>(TB->ax[0].tab)[aoff0].t;
        int cjz[HJB_MAX_C] = {0, 0, 0};
        T gpre = (T)0;
        for (int k = 0; k < P->n_cost_prefix; ++k) {
            const T x = term_value<T, D>(P->cost[k], si, cjz);
            gpre = (k == 0) ? x : (T)(gpre + x);
        }
        // ---- phase A: lane u prepares control u ------------------------------------------------------
        if (lane < nU) {
            uint32_t rb = 0;
            bool bad = false;
#pragma unroll
            for (int a = 1; a < D; ++a) {
                const DTabled::Axis &A = TB->ax[a];
                int off = A.cstride[0] * my_cj[0] + A.cstride[1] * my_cj[1] + A.cstride[2] * my_cj[2];
#pragma unroll
                for (int d = 1; d < D; ++d) off += A.sstride[d] * sl[d];
                int cl = as_global<TabEntry<T>>(A.tab)[off].cell;
                s_tw[lane * DR + (a - 1)] = as_global<TabEntry<T>>(A.tab)[off].t;
                if (a == D - 1) {
                    cl -= plane0;
                    if (cl < 0 || cl + 1 >= nplanes) { bad = true; cl = cl < 0 ? 0 : nplanes - 2; }
                }
                rb += js[a] * (uint32_t)cl;
            }
            if (bad) *P->status = 1;
            s_rb[lane] = rb;
            for (int k = 0; k < ncu; ++k) {
                const DTerm &t = P->cost[P->n_cost_prefix + k];
                const int off = t.stride[D] * my_cj[0] + t.stride[D + 1] * my_cj[1] + t.stride[D + 2] * my_cj[2];
                s_cu[lane * kLeanMaxCu + k] = as_global<T>(t.data)[off];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- phase B: the control loop ------------------------------------------------------------------
        T best = (T)0;
        int best_u = 0;
        const char *Jb = reinterpret_cast<const char *>(Jn);
        // the upper axis-0 neighbour through a second, opaque base: the two 4-byte loads of a corner pair must stay
        // two instructions - merged into one unaligned 8-byte load they are slower (measured: 9.1 vs 7.5 ms on C4)
        gptr<char> Jb1 = as_global<char>(Jb + sizeof(TJ));
        asm volatile("" : "+s"(Jb1));
        for (int u = 0; u < nU; ++u) {
            // BYTE offsets in 32 bits (J is < 4 GiB here): `uniform base + zero-extended VGPR offset + immediate`
            // is one addressing mode of global_load - no per-load 64-bit arithmetic
            const uint32_t row = (s_rb[u] + c0) * (uint32_t)sizeof(TJ);
            T v[1 << D];
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const uint32_t o = row + delta[c] * (uint32_t)sizeof(TJ);
                v[2 * c] = (T) * reinterpret_cast<const TJ *>(Jb + o);
                v[2 * c + 1] = (T) * reinterpret_cast<gptr<TJ>>(Jb1 + o);
            }
            // (packing these lerps two rows per v_pk_fma_f32 was measured slower: 7.3 vs 6.8 ms on C4 - the loaded
            // corners land in unrelated registers and have to be moved into pairs first)
#pragma unroll
            for (int j = 0; j < NR; ++j) v[j] = fma_t<T>(t0, (T)(v[2 * j + 1] - v[2 * j]), v[2 * j]);
#pragma unroll
            for (int a = 1; a < D; ++a) {
                const T ta = s_tw[u * DR + (a - 1)];
#pragma unroll
                for (int j = 0; j < (1 << (D - 1 - a)); ++j)
                    v[j] = fma_t<T>(ta, (T)(v[2 * j + 1] - v[2 * j]), v[2 * j]);
            }
            const T interp = v[0];
            T g = gpre;
            if (ncu == 1 && has_prefix) {                          // the usual shape: state terms + one control term
                g = (T)(g + s_cu[u * kLeanMaxCu]);
            } else {
                for (int k = 0; k < ncu; ++k) {
                    const T x = s_cu[u * kLeanMaxCu + k];
                    g = (!has_prefix && k == 0) ? x : (T)(g + x);
                }
            }
            const T tot = (T)(g + interp);
            if (u == 0 || tot < best) {
                best = tot;
                best_u = u;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const int j0 = s_cj[3 * best_u], j1 = s_cj[3 * best_u + 1], j2 = s_cj[3 * best_u + 2];
            const int label = C == 1 ? j0 : (C == 2 ? j0 + P->m[0] * j1 : j0 + P->m[0] * (j1 + P->m[1] * j2));
            uint32_t ls = 0, mul = 1, lj = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                ls += mul * (uint32_t)sl[a];
                lj += js[a] * (uint32_t)(a == D - 1 ? sl[a] + P->halo_lo : sl[a]);
                mul *= (uint32_t)P->n[a];
            }
            stj<T, TJ>(Jout, lj, best);
            if (idx_out) st_idx(idx_out, ls, label + P->index_base, P->idx_bytes);
        }
    }
}

}  // namespace hjb
