// kernels_lookup.h - K5: batched policy / value lookup on a grid.
// One thread per query point.  'linear' uses the sweep's canonical interpolation
// (exact cell search, t = (q-k[c])*rdx[c], fma lerps axis 0 first); 'nearest' picks,
// per axis, the nearer knot of the enclosing cell (upper knot at the midpoint).
#pragma once
#include "hjbdp_dev.h"
#include "kernels_generic.h"

namespace hjb {

struct DLookup {
    int32_t D, method;
    int32_t n[HJB_MAX_D];
    int64_t stride[HJB_MAX_D];
    const void *knots[HJB_MAX_D];
    const void *rdx[HJB_MAX_D];
};

template <typename T, int D>
__global__ void __launch_bounds__(256)
k_policy_lookup(DLookup L, const T *__restrict__ V, int64_t nq, const T *__restrict__ Q, T *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t base = 0;
        T tw[D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const T q = Q[i * D + a];
            const T *kk = static_cast<const T *>(L.knots[a]);
            const int n = L.n[a];
            int cell = find_cell<T>(kk, n, q, 0, (T)0, (T)0);
            if (L.method == HJB_LOOKUP_NEAREST) {
                if ((T)(q - kk[cell]) >= (T)(kk[cell + 1] - q)) ++cell;
                tw[a] = (T)0;
            } else {
                tw[a] = (T)((T)(q - kk[cell]) * static_cast<const T *>(L.rdx[a])[cell]);
            }
            base += L.stride[a] * cell;
        }
        if (L.method == HJB_LOOKUP_NEAREST) {
            out[i] = V[base];
            continue;
        }
        T v[1 << D];
#pragma unroll
        for (int c = 0; c < (1 << D); ++c) {
            int64_t off = base;
#pragma unroll
            for (int a = 0; a < D; ++a)
                if (c & (1 << a)) off += L.stride[a];
            v[c] = V[off];
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
#pragma unroll
            for (int j = 0; j < (1 << (D - 1 - a)); ++j)
                v[j] = fma_t<T>(tw[a], (T)(v[2 * j + 1] - v[2 * j]), v[2 * j]);
        }
        out[i] = v[0];
    }
}

}  // namespace hjb
