// stage_packed.hip - variant 2 (kernels_packed.h): two states per packed fp32 op
// One translation unit per stage-kernel family (hjbdp_launch.h): built in parallel by __graft_entry__.build().
#include "hjbdp_launch.h"
#include "kernels_packed.h"

namespace hjb {

int stage_packed(const StageArgs &a) {
    if (a.dtype != HJB_F32) return 1;
    const dim3 g(a.grid), b(a.block);
    const float *Jn = (const float *)a.Jn;
    float *Jo = (float *)a.Jo;
    switch (a.D) {
        case 1: hipLaunchKernelGGL((k_backup_packed<1>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx); break;
        case 2: hipLaunchKernelGGL((k_backup_packed<2>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx); break;
        case 3: hipLaunchKernelGGL((k_backup_packed<3>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx); break;
        case 4: hipLaunchKernelGGL((k_backup_packed<4>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx); break;
        case 5: hipLaunchKernelGGL((k_backup_packed<5>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx); break;
        case 6: hipLaunchKernelGGL((k_backup_packed<6>), g, b, a.lds, a.st, a.dp, a.dn, Jn, Jo, a.idx); break;
        default: return 1;
    }
    return 0;
}

}  // namespace hjb
