// stage_tabled.hip - variant 5 (K7, kernels_tabled.h): the general kernel on precomputed (cell, t) tables
// One translation unit per stage-kernel family (hjbdp_launch.h): built in parallel by __graft_entry__.build().
#include "hjbdp_launch.h"
#include "kernels_tabled.h"

namespace hjb {

template <typename T, typename TJ>
static int go(const StageArgs &a) {
    const dim3 g(a.grid), b(a.block);
    const TJ *Jn = (const TJ *)a.Jn;
    TJ *Jo = (TJ *)a.Jo;
    if (a.idx32) {
        switch (a.D) {
            case 1: hipLaunchKernelGGL((k_backup_tabled32<T, TJ, 1>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
            case 2: hipLaunchKernelGGL((k_backup_tabled32<T, TJ, 2>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
            case 3: hipLaunchKernelGGL((k_backup_tabled32<T, TJ, 3>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
            case 4: hipLaunchKernelGGL((k_backup_tabled32<T, TJ, 4>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
            case 5: hipLaunchKernelGGL((k_backup_tabled32<T, TJ, 5>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
            case 6: hipLaunchKernelGGL((k_backup_tabled32<T, TJ, 6>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
            default: return 1;
        }
        return 0;
    }
    switch (a.D) {
        case 1: hipLaunchKernelGGL((k_backup_tabled<T, TJ, 1>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
        case 2: hipLaunchKernelGGL((k_backup_tabled<T, TJ, 2>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
        case 3: hipLaunchKernelGGL((k_backup_tabled<T, TJ, 3>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
        case 4: hipLaunchKernelGGL((k_backup_tabled<T, TJ, 4>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
        case 5: hipLaunchKernelGGL((k_backup_tabled<T, TJ, 5>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
        case 6: hipLaunchKernelGGL((k_backup_tabled<T, TJ, 6>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx); break;
        default: return 1;
    }
    return 0;
}

// n problems of one (dtype, D) as one launch of the 32-bit form (blockIdx.y = the problem): hjb_solve_batch
template <typename T, typename TJ>
static int go_batch(const StageArgs &a, int n, const DCsBatch &hB, uint32_t mask, int parity) {
    const dim3 g(a.grid, (unsigned)n), b(a.block);
    switch (a.D) {
        case 1: hipLaunchKernelGGL((k_backup_tabled32_batch<T, TJ, 1>), g, b, 0, a.st, hB, mask, parity); break;
        case 2: hipLaunchKernelGGL((k_backup_tabled32_batch<T, TJ, 2>), g, b, 0, a.st, hB, mask, parity); break;
        case 3: hipLaunchKernelGGL((k_backup_tabled32_batch<T, TJ, 3>), g, b, 0, a.st, hB, mask, parity); break;
        case 4: hipLaunchKernelGGL((k_backup_tabled32_batch<T, TJ, 4>), g, b, 0, a.st, hB, mask, parity); break;
        default: return 1;       // (5-D / 6-D grids are not launch-bound: hjb_solve on threads of their own)
    }
    return 0;
}

int stage_tabled_batch(const StageArgs &a, int n, const DCsBatch &hB, uint32_t mask, int parity) {
    if (a.dtype == HJB_F16S) return go_batch<float, _Float16>(a, n, hB, mask, parity);
    if (a.dtype == HJB_F32) return go_batch<float, float>(a, n, hB, mask, parity);
    return go_batch<double, double>(a, n, hB, mask, parity);
}

int stage_tabled(const StageArgs &a) {
    if (a.dtype == HJB_F16S) return go<float, _Float16>(a);
    if (a.dtype == HJB_F32) return go<float, float>(a);
    return go<double, double>(a);
}

}  // namespace hjb
