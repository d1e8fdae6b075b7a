// kernels_probe.h - the reference's debug taps as a kernel (test/Dynamic_Solver.m:212-219, `checkstagesXJF`):
// for a rectangular sub-block of states and ONE control, the stage cost g(x,u), the next-state coordinate of every
// axis and J_{k+1} interpolated there - the quantities the reference copies out of its J_current_state, X_next_M*
// and J_F_next tables per stage.  Same canonical arithmetic as the stage kernels (ordered term sums, exact cell
// search, axis-0-first lerps), one thread per block state; a few hundred states at most, never on the hot path.
#pragma once
#include "hjbdp_dev.h"
#include "kernels_generic.h"

namespace hjb {

struct DProbe {
    int32_t lo[HJB_MAX_D], ext[HJB_MAX_D];   // first state index and extent per axis (global indices)
    int32_t control[HJB_MAX_C];
    int32_t pad;
    int64_t B;                               // block states
    void *g, *x_next, *j_interp;             // device outputs of ONE stage (may be null)
};

template <typename T, typename TJ, int D>
__global__ void __launch_bounds__(256)
k_probe(const DParams *__restrict__ P, DProbe pr, const TJ *__restrict__ Jn) {
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < pr.B; b += (int64_t)gridDim.x * blockDim.x) {
        int si[D];
        int64_t r = b;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            si[a] = pr.lo[a] + (int)(r % pr.ext[a]);
            r /= pr.ext[a];
        }
        const int cj[HJB_MAX_C] = {pr.control[0], pr.control[1], pr.control[2]};
        T tw[D];
        int64_t base = 0;
        bool bad = false;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const DAxis &ax = P->axis[a];
            T q = (T)0;
            for (int k = 0; k < ax.n_terms; ++k) {
                const T x = term_value<T, D>(ax.t[k], si, cj);
                q = (k == 0) ? x : (T)(q + x);
            }
            if (pr.x_next) static_cast<T *>(pr.x_next)[b + pr.B * a] = q;
            const T *kk = static_cast<const T *>(ax.knots);
            int cell = find_cell<T>(kk, ax.n, q, ax.uniform, (T)ax.x0, (T)ax.inv_h);
            tw[a] = (T)((T)(q - kk[cell]) * static_cast<const T *>(ax.rdx)[cell]);
            if (a == D - 1) {
                cell -= P->plane0;
                if (cell < 0 || cell + 1 >= P->nplanes) { bad = true; cell = cell < 0 ? 0 : P->nplanes - 2; }
            }
            base += P->jstride[a] * cell;
        }
        if (pr.g) {
            T g = (T)0;
            for (int k = 0; k < P->n_cost; ++k) {
                const T x = term_value<T, D>(P->cost[k], si, cj);
                g = (k == 0) ? x : (T)(g + x);
            }
            static_cast<T *>(pr.g)[b] = g;
        }
        if (pr.j_interp && Jn) {
            if (bad) *P->status = 1;
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
            static_cast<T *>(pr.j_interp)[b] = v[0];
        }
    }
}

}  // namespace hjb
