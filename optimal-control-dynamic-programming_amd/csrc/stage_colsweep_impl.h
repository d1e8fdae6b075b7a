// stage_colsweep_impl.h - the instantiations of k_backup_colsweep for one (J storage, group axis) (see stage_colsweep.hip)
#pragma once
#include "hjbdp_launch.h"
#include "kernels_colsweep.h"

namespace hjb {

template <typename TJ, int GAX>
static int colsweep_go(const StageArgs &a, int ng, bool fastcost, bool dpp) {
    const dim3 g(a.grid), b(a.block);
    const TJ *Jn = (const TJ *)a.Jn;
    TJ *Jo = (TJ *)a.Jo;
#define HJB_CS(NG)                                                                                                     \
    case NG:                                                                                                           \
        if (fastcost) {                                                                                                \
            if (dpp) hipLaunchKernelGGL((k_backup_colsweep<float, TJ, GAX, NG, true, true>), g, b, 0, a.st, a.dp, a.dtb, a.dcs, Jn, Jo, a.idx);   \
            else hipLaunchKernelGGL((k_backup_colsweep<float, TJ, GAX, NG, true, false>), g, b, 0, a.st, a.dp, a.dtb, a.dcs, Jn, Jo, a.idx);      \
        } else {                                                                                                       \
            if (dpp) hipLaunchKernelGGL((k_backup_colsweep<float, TJ, GAX, NG, false, true>), g, b, 0, a.st, a.dp, a.dtb, a.dcs, Jn, Jo, a.idx);  \
            else hipLaunchKernelGGL((k_backup_colsweep<float, TJ, GAX, NG, false, false>), g, b, 0, a.st, a.dp, a.dtb, a.dcs, Jn, Jo, a.idx);     \
        }                                                                                                              \
        break;
    switch (ng) {
        HJB_CS(1) HJB_CS(2) HJB_CS(3) HJB_CS(4) HJB_CS(5) HJB_CS(6)
        default: return 1;
    }
#undef HJB_CS
    return 0;
}

// the batched launch (kernels_colsweep.h: k_backup_colsweep_batch): float32 J, the usual cost shape, the one-load form
template <int GAX, bool C64>
static int colsweep_go_batch(const StageArgs &a, int n, const DCsBatch *dB, uint32_t mask, int parity, int ng) {
    const dim3 g(a.grid, (unsigned)n), b(a.block);
#define HJB_CSB(NG)                                                                                                    \
    case NG: hipLaunchKernelGGL((k_backup_colsweep_batch<float, float, GAX, NG, true, true, C64>), g, b, 0, a.st, dB, mask, parity); break;
    switch (ng) {
        HJB_CSB(1) HJB_CSB(2) HJB_CSB(3) HJB_CSB(4) HJB_CSB(5) HJB_CSB(6)
        default: return 1;
    }
#undef HJB_CSB
    return 0;
}

// cost form 2 (hjb_problem.cost_dtype == HJB_COST_F64: state terms + one control term, summed in double): units of their own
template <typename TJ, int GAX>
static int colsweep_go_c64(const StageArgs &a, int ng, bool dpp) {
    const dim3 g(a.grid), b(a.block);
    const TJ *Jn = (const TJ *)a.Jn;
    TJ *Jo = (TJ *)a.Jo;
#define HJB_CS64(NG)                                                                                                   \
    case NG:                                                                                                           \
        if (dpp) hipLaunchKernelGGL((k_backup_colsweep<float, TJ, GAX, NG, true, true, true>), g, b, 0, a.st, a.dp, a.dtb, a.dcs, Jn, Jo, a.idx);    \
        else hipLaunchKernelGGL((k_backup_colsweep<float, TJ, GAX, NG, true, false, true>), g, b, 0, a.st, a.dp, a.dtb, a.dcs, Jn, Jo, a.idx);       \
        break;
    switch (ng) {
        HJB_CS64(1) HJB_CS64(2) HJB_CS64(3) HJB_CS64(4) HJB_CS64(5) HJB_CS64(6)
        default: return 1;
    }
#undef HJB_CS64
    return 0;
}

}  // namespace hjb
