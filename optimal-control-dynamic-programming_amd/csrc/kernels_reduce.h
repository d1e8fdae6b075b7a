// kernels_reduce.h - K4: the early-stop monitor's two sums
//   fsum50 = sum(F_gI.Values(:)),  idsum50 = sum(U_Optimal_id(:))
// (pos-att/Solver_pos_att.m:273-285).  Accumulated in double with a fixed
// reduction tree (fixed grid, no atomics) so the stop decision is reproducible
// run to run.
#pragma once
#include "hjbdp_dev.h"

namespace hjb {

constexpr int kReduceBlocks = 512;
constexpr int kReduceThreads = 256;

template <typename T>
__global__ void __launch_bounds__(kReduceThreads)
k_partial_sums(const T *__restrict__ J, const int32_t *__restrict__ idx, int64_t n, double *__restrict__ partials) {
    __shared__ double sj[kReduceThreads];
    __shared__ double si[kReduceThreads];
    double aj = 0.0, ai = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kReduceThreads + threadIdx.x; i < n; i += (int64_t)kReduceBlocks * kReduceThreads) {
        aj += (double)J[i];
        ai += idx ? (double)idx[i] : 0.0;
    }
    sj[threadIdx.x] = aj;
    si[threadIdx.x] = ai;
    __syncthreads();
    for (int s = kReduceThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sj[threadIdx.x] += sj[threadIdx.x + s];
            si[threadIdx.x] += si[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = sj[0];
        partials[2 * blockIdx.x + 1] = si[0];
    }
}

__global__ void __launch_bounds__(kReduceThreads)
k_final_sums(const double *__restrict__ partials, double *__restrict__ sums) {
    __shared__ double sj[kReduceThreads];
    __shared__ double si[kReduceThreads];
    double aj = 0.0, ai = 0.0;
    for (int b = threadIdx.x; b < kReduceBlocks; b += kReduceThreads) {
        aj += partials[2 * b];
        ai += partials[2 * b + 1];
    }
    sj[threadIdx.x] = aj;
    si[threadIdx.x] = ai;
    __syncthreads();
    for (int s = kReduceThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sj[threadIdx.x] += sj[threadIdx.x + s];
            si[threadIdx.x] += si[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[0] = sj[0];
        sums[1] = si[0];
    }
}

inline int launch_monitor_sums(int dtype, const void *J, const int32_t *idx, int64_t n, double *partials, double *sums,
                               hipStream_t st) {
    if (dtype == HJB_F16S)
        hipLaunchKernelGGL((k_partial_sums<_Float16>), dim3(kReduceBlocks), dim3(kReduceThreads), 0, st, (const _Float16 *)J, idx, n, partials);
    else if (dtype == HJB_F32)
        hipLaunchKernelGGL((k_partial_sums<float>), dim3(kReduceBlocks), dim3(kReduceThreads), 0, st, (const float *)J, idx, n, partials);
    else
        hipLaunchKernelGGL((k_partial_sums<double>), dim3(kReduceBlocks), dim3(kReduceThreads), 0, st, (const double *)J, idx, n, partials);
    hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(kReduceThreads), 0, st, partials, sums);
    return hipGetLastError() == hipSuccess ? HJB_OK : HJB_E_DEVICE;
}

}  // namespace hjb
