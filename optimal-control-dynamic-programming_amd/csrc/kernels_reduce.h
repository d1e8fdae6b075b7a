// kernels_reduce.h - K4: the early-stop monitor's two sums
//   fsum50 = sum(F_gI.Values(:)),  idsum50 = sum(U_Optimal_id(:))
// (pos-att/Solver_pos_att.m:273-285).  A fixed reduction tree (fixed grid, no atomics), so the stop decision is
// reproducible run to run.  The tree, which oracle/hjb_oracle.c restates element for element:
//   accumulator a = i mod (kReduceBlocks * kReduceThreads) takes elements i = a, a + 131072, ... in ascending order;
//   inside a block of kReduceThreads accumulators: pairwise halving (s = 128, 64, ... 1: acc[t] += acc[t + s]);
//   the block sums b = t, t + 256 are added in ascending order by accumulator t, then the same pairwise halving.
// The sum of J is carried in TA = double (exact enough to be order-free: the library's default) or in TA = float
// (hjb_solve_opts.monitor_single: MATLAB's sum() of a single array is a single-precision sum whose own order is not
// documented - this tree is the order we state); the sum of the labels is integer-valued and always exact in double.
#pragma once
#include "hjbdp_dev.h"

namespace hjb {

constexpr int kReduceBlocks = 512;
constexpr int kReduceThreads = 256;

__device__ __forceinline__ double ld_idx(const void *__restrict__ base, int64_t i, int32_t bytes) {
    if (bytes == 4) return (double)((const int32_t *)base)[i];
    if (bytes == 1) return (double)((const uint8_t *)base)[i];
    return (double)((const uint16_t *)base)[i];
}

template <typename T, typename TA>
__global__ void __launch_bounds__(kReduceThreads)
k_partial_sums(const T *__restrict__ J, const void *__restrict__ idx, int32_t idx_bytes, int64_t n, double *__restrict__ partials) {
    __shared__ TA sj[kReduceThreads];
    __shared__ double si[kReduceThreads];
    TA aj = (TA)0;
    double ai = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kReduceThreads + threadIdx.x; i < n; i += (int64_t)kReduceBlocks * kReduceThreads) {
        aj = (TA)(aj + (TA)J[i]);
        ai += idx ? ld_idx(idx, i, idx_bytes) : 0.0;
    }
    sj[threadIdx.x] = aj;
    si[threadIdx.x] = ai;
    __syncthreads();
    for (int s = kReduceThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sj[threadIdx.x] = (TA)(sj[threadIdx.x] + sj[threadIdx.x + s]);
            si[threadIdx.x] += si[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (double)sj[0];          // a float partial is exact in the double slot
        partials[2 * blockIdx.x + 1] = si[0];
    }
}

template <typename TA>
__global__ void __launch_bounds__(kReduceThreads)
k_final_sums(const double *__restrict__ partials, double *__restrict__ sums) {
    __shared__ TA sj[kReduceThreads];
    __shared__ double si[kReduceThreads];
    TA aj = (TA)0;
    double ai = 0.0;
    for (int b = threadIdx.x; b < kReduceBlocks; b += kReduceThreads) {
        aj = (TA)(aj + (TA)partials[2 * b]);
        ai += partials[2 * b + 1];
    }
    sj[threadIdx.x] = aj;
    si[threadIdx.x] = ai;
    __syncthreads();
    for (int s = kReduceThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sj[threadIdx.x] = (TA)(sj[threadIdx.x] + sj[threadIdx.x + s]);
            si[threadIdx.x] += si[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[0] = (double)sj[0];
        sums[1] = si[0];
    }
}

inline int launch_monitor_sums(int dtype, bool single_sum, const void *J, const void *idx, int32_t idx_bytes, int64_t n,
                               double *partials, double *sums, hipStream_t st) {
    const dim3 g(kReduceBlocks), b(kReduceThreads);
    if (dtype == HJB_F16S) {
        if (single_sum) hipLaunchKernelGGL((k_partial_sums<_Float16, float>), g, b, 0, st, (const _Float16 *)J, idx, idx_bytes, n, partials);
        else hipLaunchKernelGGL((k_partial_sums<_Float16, double>), g, b, 0, st, (const _Float16 *)J, idx, idx_bytes, n, partials);
    } else if (dtype == HJB_F32) {
        if (single_sum) hipLaunchKernelGGL((k_partial_sums<float, float>), g, b, 0, st, (const float *)J, idx, idx_bytes, n, partials);
        else hipLaunchKernelGGL((k_partial_sums<float, double>), g, b, 0, st, (const float *)J, idx, idx_bytes, n, partials);
    } else {
        hipLaunchKernelGGL((k_partial_sums<double, double>), g, b, 0, st, (const double *)J, idx, idx_bytes, n, partials);
        single_sum = false;                                  // a double J is summed in double (MATLAB does the same)
    }
    if (single_sum) hipLaunchKernelGGL((k_final_sums<float>), dim3(1), b, 0, st, partials, sums);
    else hipLaunchKernelGGL((k_final_sums<double>), dim3(1), b, 0, st, partials, sums);
    return hipGetLastError() == hipSuccess ? HJB_OK : HJB_E_DEVICE;
}

}  // namespace hjb
