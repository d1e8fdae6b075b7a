// kernels_ctrlsplit.h - variant 3: control-split stage kernel.
//
// For grids with FEW states and MANY controls (the Kirk example: 100x100 states x
// 1000 controls, test/Dynamic_Solver.m:53-63) one thread per state leaves most of
// the 256 CUs idle.  Here one 64-lane wavefront owns one state: the controls are
// striped across the lanes (lane l evaluates visiting indices l, l+64, ...), each
// lane keeps its first minimum with the strict '<' rule, and the wave combines
// the 64 (value, visiting index) pairs with a shuffle butterfly that prefers the
// smaller value and, on equal values, the smaller visiting index - which is the
// global first minimum in visiting order, i.e. MATLAB's min / the cascade rule.
// The whole J_{k+1} grid is staged in LDS when it fits (Kirk: 100*100*4 B = 40 KB),
// so the 2^D-corner gathers never leave the CU.  Per-backup arithmetic is the
// canonical order of the generic kernel: results are bit-identical.
#pragma once
#include "hjbdp_dev.h"
#include "kernels_generic.h"

namespace hjb {

template <typename T, int D, bool J_IN_LDS>
__global__ void __launch_bounds__(1024)
k_backup_ctrlsplit(const DParams *__restrict__ P, const T *__restrict__ Jn, T *__restrict__ Jout,
                   void *__restrict__ idx_out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *s_J = reinterpret_cast<T *>(smem_raw);
    const int C = P->C;
    const int64_t n_owned = P->n_owned;
    const int nU = (int)P->nU;
    if constexpr (J_IN_LDS) {
        const int je = (int)(P->jstride[D - 1] * P->nplanes);
        for (int i = threadIdx.x; i < je; i += blockDim.x) s_J[i] = Jn[i];
        __syncthreads();
    }
    const T *Jsrc = J_IN_LDS ? s_J : Jn;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;

    for (int64_t ls = (int64_t)blockIdx.x * waves_per_block + wave; ls < n_owned;
         ls += (int64_t)gridDim.x * waves_per_block) {
        int si[D];
        {
            int64_t r = ls;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                int na = P->n[a];
                si[a] = (int)(r % na);
                r /= na;
            }
            si[D - 1] += P->slab_begin;
        }
        int cz[HJB_MAX_C] = {0, 0, 0};
        T qpre[D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const DAxis &ax = P->axis[a];
            T q = (T)0;
            for (int k = 0; k < ax.n_prefix; ++k) {
                T x = term_value<T, D>(ax.t[k], si, cz);
                q = (k == 0) ? x : (T)(q + x);
            }
            qpre[a] = q;
        }
        T gpre = (T)0;
        for (int k = 0; k < P->n_cost_prefix; ++k) {
            T x = term_value<T, D>(P->cost[k], si, cz);
            gpre = (k == 0) ? x : (T)(gpre + x);
        }
        T best = (T)INFINITY;
        int best_u = 0x7fffffff;
        bool have = false, first_nan = false;
        for (int u = lane; u < nU; u += 64) {
            // visiting index (control dim 0 slowest) -> per-dim control indices
            int cj[HJB_MAX_C] = {0, 0, 0};
            if (C == 1) {
                cj[0] = u;
            } else if (C == 2) {
                cj[1] = u % P->m[1];
                cj[0] = u / P->m[1];
            } else {
                cj[2] = u % P->m[2];
                const int r2 = u / P->m[2];
                cj[1] = r2 % P->m[1];
                cj[0] = r2 / P->m[1];
            }
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
                v[c] = Jsrc[off];
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
            const T tot = (T)(g + v[0]);
            // A NaN total (inf - inf in a runaway sweep) is never a candidate - except that of control 0, which the
            // sequential rule `first || tot < best` of the reference loop takes as the result (checked after the reduction):
            // a lane that HELD a NaN would answer every comparison of the butterfly with "no" and hide its whole subtree.
            if (u == 0 && tot != tot) first_nan = true;
            if (tot == tot && (!have || tot < best)) {
                best = tot;
                best_u = u;
                have = true;
            }
        }
        // wave butterfly: smaller value wins; equal values -> smaller visiting index (first minimum)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const T ov = __shfl_xor(best, off, 64);
            const int ou = __shfl_xor(best_u, off, 64);
            if (ov < best || (ov == best && ou < best_u)) {
                best = ov;
                best_u = ou;
            }
        }
        if (__any(first_nan)) {                   // control 0 (lane 0's first) was NaN: the sequential rule keeps it
            best = (T)NAN;
            best_u = 0;
        } else if (best_u == 0x7fffffff) {        // every other total NaN, none kept: control 0's own (non-NaN) total won
            best_u = 0;
        }
        if (lane == 0) {
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
            Jout[in_plane + P->inner * (pl + P->halo_lo)] = best;
            if (idx_out) st_idx(idx_out, ls, (int32_t)(label + P->index_base), P->idx_bytes);
        }
    }
}

}  // namespace hjb
