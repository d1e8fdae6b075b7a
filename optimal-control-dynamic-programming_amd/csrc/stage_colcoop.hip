// stage_colcoop.hip - variant 7, cooperative form (K13, kernels_colcoop.h): corner rows of eight neighbouring columns
// staged once through LDS.  Option "cs_coop", off by default (measured slower on C4).
#include "hjbdp_launch.h"
#include "kernels_colsweep.h"
#include "kernels_colcoop.h"

namespace hjb {

template <typename TJ>
static int go(const StageArgs &a, int gax, int ng, bool fastcost) {
    constexpr int EPL = 16 / (int)sizeof(TJ);
    const dim3 g(a.grid), b(kCcW * 64);
    const TJ *Jn = (const TJ *)a.Jn;
    TJ *Jo = (TJ *)a.Jo;
#define HJB_CC2(NG, FC)                                                                                                \
    do {                                                                                                               \
        if (gax == 3) hipLaunchKernelGGL((k_backup_colcoop<float, TJ, 3, NG, FC, EPL>), g, b, 0, a.st, a.dp, a.dtb, a.dcs, Jn, Jo, a.idx); \
        else hipLaunchKernelGGL((k_backup_colcoop<float, TJ, 2, NG, FC, EPL>), g, b, 0, a.st, a.dp, a.dtb, a.dcs, Jn, Jo, a.idx);          \
    } while (0)
#define HJB_CC(NG)                                                                                                     \
    case NG:                                                                                                           \
        if (fastcost) HJB_CC2(NG, true); else HJB_CC2(NG, false);                                                      \
        break;
    switch (ng) {
        HJB_CC(1) HJB_CC(2) HJB_CC(3) HJB_CC(4) HJB_CC(5)
        default: return 1;
    }
#undef HJB_CC2
#undef HJB_CC
    return 0;
}

int stage_colcoop(const StageArgs &a, int gax, int ng, bool fastcost) {
    if (a.dtype == HJB_F32) return go<float>(a, gax, ng, fastcost);
    if (a.dtype == HJB_F16S) return go<_Float16>(a, gax, ng, fastcost);
    return 1;
}

}  // namespace hjb
