// stage_uniwin_impl.h - the instantiations of k_backup_uniwin for one J storage type (see stage_uniwin.hip)
#pragma once
#include "hjbdp_launch.h"
#include "kernels_uniwin.h"

namespace hjb {

template <typename TJ>
static const void *uniwin_kernel(int D, bool model) {
    if (model) return D == 6 ? (const void *)k_backup_uniwin<TJ, 6, true> : nullptr;
    switch (D) {
        case 4: return (const void *)k_backup_uniwin<TJ, 4, false>;
        case 5: return (const void *)k_backup_uniwin<TJ, 5, false>;
        case 6: return (const void *)k_backup_uniwin<TJ, 6, false>;
        default: return nullptr;
    }
}

template <typename TJ>
static int uniwin_go(const StageArgs &a, bool model) {
    const dim3 g(a.grid), b(a.block);
    const TJ *Jn = (const TJ *)a.Jn;
    TJ *Jo = (TJ *)a.Jo;
    if (model) {
        if (a.D != 6) return 1;
        hipLaunchKernelGGL((k_backup_uniwin<TJ, 6, true>), g, b, a.lds, a.st, a.dp, a.dn, a.duw, Jn, Jo, a.idx);
        return 0;
    }
    switch (a.D) {
        case 4: hipLaunchKernelGGL((k_backup_uniwin<TJ, 4, false>), g, b, a.lds, a.st, a.dp, a.dn, a.duw, Jn, Jo, a.idx); break;
        case 5: hipLaunchKernelGGL((k_backup_uniwin<TJ, 5, false>), g, b, a.lds, a.st, a.dp, a.dn, a.duw, Jn, Jo, a.idx); break;
        case 6: hipLaunchKernelGGL((k_backup_uniwin<TJ, 6, false>), g, b, a.lds, a.st, a.dp, a.dn, a.duw, Jn, Jo, a.idx); break;
        default: return 1;
    }
    return 0;
}

template <typename TJ>
static int uniwin_occupancy_t(int D, bool model, size_t lds) {
    const void *k = uniwin_kernel<TJ>(D, model);
    if (!k) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, lds) != hipSuccess) return 0;
    return n;
}

}  // namespace hjb
