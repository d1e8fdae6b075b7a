// kernels_generic.h - variant 0: the fully general stage kernel.
//
// One thread per owned state; the thread walks every control (control dim 0
// slowest = the reference's cascade min, Solver_attitude.m:400-409), evaluates
// the ordered broadcast-term sums for x_next and g on the fly, finds the cell by
// exact search, gathers the 2^D corners of J_{k+1} straight from global memory
// (L1/L2/MALL resident for every reference-sized grid) and keeps the first
// minimum.  Arithmetic is the canonical order of oracle/hjb_oracle.c, so results
// are bit-identical to the CPU twin.  Any D <= 6, C <= 3, uniform or non-uniform
// knots, slabs with halos, float16 J storage.  The fast kernels (kernels_nested.h,
// kernels_packed*.h, kernels_ctrlsplit.h, kernels_tabled.h) cover the shapes that matter
// for throughput; this one is the safety net and the parity anchor.
#pragma once
#include "hjbdp_dev.h"

namespace hjb {

template <typename T> __device__ __forceinline__ T fma_t(T a, T b, T c);
template <> __device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_t<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

// clamp(upper_bound(k, q) - 1, 0, n-2), exact for any strictly increasing knots.
template <typename T>
__device__ __forceinline__ int find_cell(const T *__restrict__ k, int n, T q, int uniform, T x0, T inv_h) {
    int i;
    if (uniform) {
        T f = (q - x0) * inv_h;
        T hi = (T)(n - 2);
        f = f > (T)0 ? f : (T)0;
        f = f < hi ? f : hi;
        i = (int)f;
        while (i > 0 && q < k[i]) --i;
        while (i < n - 2 && q >= k[i + 1]) ++i;
    } else {
        int lo = 0, hi = n - 1;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (k[mid] <= q) lo = mid; else hi = mid;
        }
        i = lo;
    }
    return i;
}

template <typename T, int D>
__device__ __forceinline__ T term_value(const DTerm &t, const int (&si)[D], const int (&cj)[HJB_MAX_C]) {
    int64_t off = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) off += (int64_t)t.stride[a] * si[a];
#pragma unroll
    for (int c = 0; c < HJB_MAX_C; ++c) off += (int64_t)t.stride[D + c] * cj[c];
    return static_cast<const T *>(t.data)[off];
}

template <typename T, typename TJ, int D>
__global__ void __launch_bounds__(256)
k_backup_generic(const DParams *__restrict__ P, const TJ *__restrict__ Jn, TJ *__restrict__ Jout,
                 void *__restrict__ idx_out) {
    const int C = P->C;
    const int64_t n_owned = P->n_owned;
    const int64_t nU = P->nU;
    for (int64_t ls = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; ls < n_owned;
         ls += (int64_t)gridDim.x * blockDim.x) {
        int si[D];
        {
            int64_t r = ls;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                int na = P->n[a];
                si[a] = (int)(r % na);
                r /= na;
            }
            si[D - 1] += P->slab_begin;  // tables are indexed by GLOBAL grid indices
        }
        int cj[HJB_MAX_C] = {0, 0, 0};
        T qpre[D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const DAxis &ax = P->axis[a];
            T q = (T)0;
            for (int k = 0; k < ax.n_prefix; ++k) {
                T x = term_value<T, D>(ax.t[k], si, cj);
                q = (k == 0) ? x : (T)(q + x);
            }
            qpre[a] = q;
        }
        T gpre = (T)0;
        for (int k = 0; k < P->n_cost_prefix; ++k) {
            T x = term_value<T, D>(P->cost[k], si, cj);
            gpre = (k == 0) ? x : (T)(gpre + x);
        }

        T best = (T)0;
        int64_t best_u = 0;
        for (int64_t u = 0; u < nU; ++u) {
            T tw[D];
            int64_t base = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const DAxis &ax = P->axis[a];
                T q = qpre[a];
                for (int k = ax.n_prefix; k < ax.n_terms; ++k) {
                    T x = term_value<T, D>(ax.t[k], si, cj);
                    q = (k == 0) ? x : (T)(q + x);
                }
                const T *kk = static_cast<const T *>(ax.knots);
                int cell = find_cell<T>(kk, ax.n, q, ax.uniform, (T)ax.x0, (T)ax.inv_h);
                tw[a] = (T)((T)(q - kk[cell]) * static_cast<const T *>(ax.rdx)[cell]);
                if (a == D - 1) {
                    cell -= P->plane0;
                    if (cell < 0 || cell + 1 >= P->nplanes) {
                        *P->status = 1;
                        cell = cell < 0 ? 0 : P->nplanes - 2;
                    }
                }
                base += P->jstride[a] * cell;
            }
            T v[1 << D];
#pragma unroll
            for (int c = 0; c < (1 << D); ++c) {
                int64_t off = base;
#pragma unroll
                for (int a = 0; a < D; ++a)
                    if (c & (1 << a)) off += P->jstride[a];
                v[c] = ldj<T, TJ>(Jn, off);
            }
#pragma unroll
            for (int a = 0; a < D; ++a) {
#pragma unroll
                for (int j = 0; j < (1 << (D - 1 - a)); ++j)
                    v[j] = fma_t<T>(tw[a], (T)(v[2 * j + 1] - v[2 * j]), v[2 * j]);
            }
            T g = gpre;
            for (int k = P->n_cost_prefix; k < P->n_cost; ++k) {
                T x = term_value<T, D>(P->cost[k], si, cj);
                g = (k == 0) ? x : (T)(g + x);
            }
            T tot = (T)(g + v[0]);
            if (u == 0 || tot < best) {
                best = tot;
                best_u = u;
            }
            // next control: last control dim fastest, dim 0 slowest
            if (C == 1) {
                ++cj[0];
            } else if (C == 2) {
                if (++cj[1] == P->m[1]) { cj[1] = 0; ++cj[0]; }
            } else {
                if (++cj[2] == P->m[2]) {
                    cj[2] = 0;
                    if (++cj[1] == P->m[1]) { cj[1] = 0; ++cj[0]; }
                }
            }
        }
        // visiting index (dim 0 slowest) -> column-major label (dim 0 fastest)
        int64_t label;
        if (C == 1) {
            label = best_u;
        } else if (C == 2) {
            int64_t j1 = best_u % P->m[1], j0 = best_u / P->m[1];
            label = j0 + (int64_t)P->m[0] * j1;
        } else {
            int64_t j2 = best_u % P->m[2];
            int64_t rr = best_u / P->m[2];
            int64_t j1 = rr % P->m[1], j0 = rr / P->m[1];
            label = j0 + (int64_t)P->m[0] * (j1 + (int64_t)P->m[1] * j2);
        }
        const int64_t in_plane = ls % P->inner, pl = ls / P->inner;
        stj<T, TJ>(Jout, in_plane + P->inner * (pl + P->halo_lo), best);
        if (idx_out) st_idx(idx_out, ls, (int32_t)(label + P->index_base), P->idx_bytes);
    }
}

// Stage-invariant per-axis (cell, weight) table over the axis' broadcast domain
// (used by variant 2).  One thread per domain entry; same canonical arithmetic as
// the stage kernels: q = ordered term sum, exact cell search, t = (q-k[c])*rdx[c].
// dom_size[d] = grid size of dim d if d is in the domain, else 1; entries are laid
// out column-major over the domain dims in increasing dim order.
template <typename T, int D>
__global__ void __launch_bounds__(256)
k_prep_axis_table(const DParams *__restrict__ P, int a, const int32_t *__restrict__ dom_size, int64_t n_entries,
                  int2 *__restrict__ out) {
    const DAxis &ax = P->axis[a];
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_entries;
         e += (int64_t)gridDim.x * blockDim.x) {
        int si[D];
        int cj[HJB_MAX_C] = {0, 0, 0};
        int64_t r = e;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int sz = dom_size[d];
            si[d] = (int)(r % sz);
            r /= sz;
        }
#pragma unroll
        for (int c = 0; c < HJB_MAX_C; ++c) {
            const int sz = dom_size[D + c];
            cj[c] = (int)(r % sz);
            r /= sz;
        }
        si[D - 1] += P->slab_begin;   // term tables are indexed by GLOBAL grid indices; the domain covers owned planes
        T q = (T)0;
        for (int k = 0; k < ax.n_terms; ++k) {
            T x = term_value<T, D>(ax.t[k], si, cj);
            q = (k == 0) ? x : (T)(q + x);
        }
        const T *kk = static_cast<const T *>(ax.knots);
        const int cell = find_cell<T>(kk, ax.n, q, ax.uniform, (T)ax.x0, (T)ax.inv_h);
        const float t = (float)((T)((T)(q - kk[cell]) * static_cast<const T *>(ax.rdx)[cell]));
        out[e] = make_int2(cell, __float_as_int(t));
    }
}

}  // namespace hjb
