// stage_packed2_impl.h - the instantiations of k_backup_packed2 for one J storage type (see stage_packed2.hip)
#pragma once
#include "hjbdp_launch.h"
#include "kernels_packed2.h"

namespace hjb {

// K3's modes come in two families compiled apart: the plain / C2 modes (0, 1, 4) here, the window modes (2, 3, 5, 6) in
// stage_packed2w_*.hip WITHOUT the SLP vectoriser (-fno-slp-vectorize, __graft_entry__.UNIT_FLAGS): left on, it re-packs the
// scalar lerps of the window modes into v_pk_* with four register moves per pair (14 more VGPRs, 1.8 % slower on 24^6),
// while the C2 modes are 2 % faster with it (profiles/r04_k3_experiments.log).
template <typename TJ>
static int packed2_go(const StageArgs &a, int mode) {
    const dim3 g(a.grid), b(a.block);
    const TJ *Jn = (const TJ *)a.Jn;
    TJ *Jo = (TJ *)a.Jo;
#define HJB_PACKED2(DD)                                                                                              \
    case DD:                                                                                                         \
        if (DD == 3 && mode == 1)                                                                                    \
            hipLaunchKernelGGL((k_backup_packed2<TJ, 3, 1>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx);           \
        else if (DD == 3 && mode == 4)                                                                               \
            hipLaunchKernelGGL((k_backup_packed2<TJ, 3, 4>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx);           \
        else if (mode == 0)                                                                                          \
            hipLaunchKernelGGL((k_backup_packed2<TJ, DD, 0>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx);          \
        else                                                                                                         \
            return 1;                                                                                                \
        break;
    switch (a.D) {
        HJB_PACKED2(1) HJB_PACKED2(2) HJB_PACKED2(3) HJB_PACKED2(4) HJB_PACKED2(5) HJB_PACKED2(6)
        default: return 1;
    }
#undef HJB_PACKED2
    return 0;
}

template <typename TJ>
static int packed2_go_window(const StageArgs &a, int mode) {
    const dim3 g(a.grid), b(a.block);
    const TJ *Jn = (const TJ *)a.Jn;
    TJ *Jo = (TJ *)a.Jo;
#define HJB_PACKED2W(DD)                                                                                             \
    case DD:                                                                                                         \
        if (DD == 6 && mode == 3)                                                                                    \
            hipLaunchKernelGGL((k_backup_packed2<TJ, 6, 3>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx);           \
        else if (mode == 5)                                                                                          \
            hipLaunchKernelGGL((k_backup_packed2<TJ, DD, 5>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx);          \
        else if (DD == 6 && mode == 6)                                                                               \
            hipLaunchKernelGGL((k_backup_packed2<TJ, 6, 6>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx);           \
        else if (mode == 2)                                                                                          \
            hipLaunchKernelGGL((k_backup_packed2<TJ, DD, 2>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx);          \
        else                                                                                                         \
            return 1;                                                                                                \
        break;
    switch (a.D) {
        HJB_PACKED2W(4) HJB_PACKED2W(5) HJB_PACKED2W(6)
        default: return 1;
    }
#undef HJB_PACKED2W
    return 0;
}

}  // namespace hjb
