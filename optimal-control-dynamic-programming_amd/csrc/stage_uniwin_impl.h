// stage_uniwin_impl.h - the instantiations of k_backup_uniwin for one J storage type (see stage_uniwin.hip)
#pragma once
#include "hjbdp_launch.h"
#include "kernels_uniwin.h"

namespace hjb {

template <typename TJ, int BLOCK>
static const void *uniwin_kernel_b(int D, bool model) {
    if (model) return D == 6 ? (const void *)k_backup_uniwin<TJ, 6, true, BLOCK> : nullptr;
    switch (D) {
        case 4: return (const void *)k_backup_uniwin<TJ, 4, false, BLOCK>;
        case 5: return (const void *)k_backup_uniwin<TJ, 5, false, BLOCK>;
        case 6: return (const void *)k_backup_uniwin<TJ, 6, false, BLOCK>;
        default: return nullptr;
    }
}

template <typename TJ, int BLOCK>
static int uniwin_go_b(const StageArgs &a, bool model) {
    const dim3 g(a.grid), b(BLOCK);
    const TJ *Jn = (const TJ *)a.Jn;
    TJ *Jo = (TJ *)a.Jo;
    if (model) {
        if (a.D != 6) return 1;
        hipLaunchKernelGGL((k_backup_uniwin<TJ, 6, true, BLOCK>), g, b, a.lds, a.st, a.dp, a.dn, a.duw, Jn, Jo, a.idx);
        return 0;
    }
    switch (a.D) {
        case 4: hipLaunchKernelGGL((k_backup_uniwin<TJ, 4, false, BLOCK>), g, b, a.lds, a.st, a.dp, a.dn, a.duw, Jn, Jo, a.idx); break;
        case 5: hipLaunchKernelGGL((k_backup_uniwin<TJ, 5, false, BLOCK>), g, b, a.lds, a.st, a.dp, a.dn, a.duw, Jn, Jo, a.idx); break;
        case 6: hipLaunchKernelGGL((k_backup_uniwin<TJ, 6, false, BLOCK>), g, b, a.lds, a.st, a.dp, a.dn, a.duw, Jn, Jo, a.idx); break;
        default: return 1;
    }
    return 0;
}

// a.block: 256 or 64 states per workgroup (DUniwin::block says the same to the kernel)
template <typename TJ>
static int uniwin_go(const StageArgs &a, bool model) {
    return a.block == 64 ? uniwin_go_b<TJ, 64>(a, model) : uniwin_go_b<TJ, 256>(a, model);
}

template <typename TJ>
static int uniwin_occupancy_t(int D, bool model, int block, size_t lds) {
    const void *k = block == 64 ? uniwin_kernel_b<TJ, 64>(D, model) : uniwin_kernel_b<TJ, 256>(D, model);
    if (!k) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, block, lds) != hipSuccess) return 0;
    return n;
}

}  // namespace hjb
