// stage_nested.hip - variant 1 (K2, kernels_nested.h): control-nested sweep
// One translation unit per stage-kernel family (hjbdp_launch.h): built in parallel by __graft_entry__.build().
#include "hjbdp_launch.h"
#include "kernels_nested.h"

namespace hjb {

template <typename T>
static int go(const StageArgs &a, bool fast) {
    const dim3 g(a.grid), b(a.block);
    const T *Jn = (const T *)a.Jn;
    T *Jo = (T *)a.Jo;
#define HJB_NESTED(DD)                                                                                          \
    case DD:                                                                                                    \
        if (fast) hipLaunchKernelGGL((k_backup_nested<T, DD, true>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx); \
        else hipLaunchKernelGGL((k_backup_nested<T, DD, false>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx);  \
        break;
    switch (a.D) {
        HJB_NESTED(1) HJB_NESTED(2) HJB_NESTED(3) HJB_NESTED(4) HJB_NESTED(5) HJB_NESTED(6)
        default: return 1;
    }
#undef HJB_NESTED
    return 0;
}

int stage_nested(const StageArgs &a, bool fast) {
    if (a.dtype == HJB_F32) return go<float>(a, fast);
    if (a.dtype == HJB_F64) return go<double>(a, fast);
    return 1;                        // J stored in the arithmetic type only
}

}  // namespace hjb
