// stage_rowwise.hip - variant 6 (K8, kernels_rowwise.h): one wavefront per grid row
// One translation unit per stage-kernel family (hjbdp_launch.h): built in parallel by __graft_entry__.build().
#include "hjbdp_launch.h"
#include "kernels_rowwise.h"

namespace hjb {

template <typename T, typename TJ>
static int go(const StageArgs &a, bool lean) {
    const dim3 g(a.grid), b(a.block);
    const TJ *Jn = (const TJ *)a.Jn;
    TJ *Jo = (TJ *)a.Jo;
#define HJB_ROW(DD)                                                                                              \
    case DD:                                                                                                     \
        if (lean) hipLaunchKernelGGL((k_backup_rowlean<T, TJ, DD>), g, b, a.lds, a.st, a.dp, a.dtb, Jn, Jo, a.idx); \
        else hipLaunchKernelGGL((k_backup_rowwise<T, TJ, DD>), g, b, 0, a.st, a.dp, a.dtb, Jn, Jo, a.idx);        \
        break;
    switch (a.D) {
        HJB_ROW(2) HJB_ROW(3) HJB_ROW(4) HJB_ROW(5) HJB_ROW(6)
        default: return 1;
    }
#undef HJB_ROW
    return 0;
}

int stage_rowwise(const StageArgs &a, bool lean) {
    if (a.dtype == HJB_F16S) return go<float, _Float16>(a, lean);
    if (a.dtype == HJB_F32) return go<float, float>(a, lean);
    return go<double, double>(a, lean);
}

}  // namespace hjb
