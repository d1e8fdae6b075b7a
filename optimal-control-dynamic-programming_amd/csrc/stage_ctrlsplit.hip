// stage_ctrlsplit.hip - variant 3 (K5, kernels_ctrlsplit.h): one wavefront per state, controls across the lanes
// One translation unit per stage-kernel family (hjbdp_launch.h): built in parallel by __graft_entry__.build().
#include "hjbdp_launch.h"
#include "kernels_ctrlsplit.h"

namespace hjb {

template <typename T>
static int go(const StageArgs &a, bool j_in_lds) {
    const dim3 g(a.grid), b(a.block);
    const T *Jn = (const T *)a.Jn;
    T *Jo = (T *)a.Jo;
#define HJB_SPLIT(DD)                                                                                            \
    case DD:                                                                                                     \
        if (j_in_lds) hipLaunchKernelGGL((k_backup_ctrlsplit<T, DD, true>), g, b, a.lds, a.st, a.dp, Jn, Jo, a.idx); \
        else hipLaunchKernelGGL((k_backup_ctrlsplit<T, DD, false>), g, b, 0, a.st, a.dp, Jn, Jo, a.idx);          \
        break;
    switch (a.D) {
        HJB_SPLIT(1) HJB_SPLIT(2) HJB_SPLIT(3) HJB_SPLIT(4) HJB_SPLIT(5) HJB_SPLIT(6)
        default: return 1;
    }
#undef HJB_SPLIT
    return 0;
}

int stage_ctrlsplit(const StageArgs &a, bool j_in_lds) {
    if (a.dtype == HJB_F32) return go<float>(a, j_in_lds);
    if (a.dtype == HJB_F64) return go<double>(a, j_in_lds);
    return 1;
}

}  // namespace hjb
