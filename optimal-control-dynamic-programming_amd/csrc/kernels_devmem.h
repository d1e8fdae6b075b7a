// kernels_devmem.h - device-buffer helpers behind hjb_device_fill_separable / hjb_device_gather (include/hjbdp.h):
// what a host needs to drive grids that never exist in host memory (C3: 51^6 states) through hjb_backup_stage_device.
#pragma once
#include "hjbdp_dev.h"

namespace hjb {

struct DSeparable {
    const void *v[HJB_MAX_D];
    int32_t n[HJB_MAX_D];
    int32_t D, pad;
    int64_t total;
};

// J[s] = ((v0[i0] + v1[i1]) + v2[i2]) + ... : one add of the arithmetic type per axis, axis 0 first; stored as TJ.
template <typename T, typename TJ>
__global__ void __launch_bounds__(256)
k_fill_separable(DSeparable S, TJ *__restrict__ J) {
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < S.total; s += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = s;
        T acc = (T)0;
        for (int a = 0; a < S.D; ++a) {
            const int i = (int)(r % S.n[a]);
            r /= S.n[a];
            const T x = static_cast<const T *>(S.v[a])[i];
            acc = a == 0 ? x : (T)(acc + x);
        }
        J[s] = (TJ)acc;
    }
}

__global__ void __launch_bounds__(256)
k_gather_bytes(const unsigned char *__restrict__ src, int32_t eb, const int64_t *__restrict__ sel, int64_t n, unsigned char *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = sel[i];
        for (int b = 0; b < eb; ++b) out[i * eb + b] = src[s * eb + b];
    }
}

}  // namespace hjb
