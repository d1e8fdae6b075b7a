// stage_uniwin.hip - variant 4 modes 7 / 8 (K15, kernels_uniwin.h): the window kernel for chunks that share their rate axes.
// The two J storage types are compiled in translation units of their own (stage_uniwin_f32.hip, stage_uniwin_f16.hip); the plan
// kernel lives here.
#include "hjbdp_launch.h"
#include "kernels_uniwin.h"

namespace hjb {

int stage_uniwin_f32(const StageArgs &a, bool model);
int stage_uniwin_f16(const StageArgs &a, bool model);
int uniwin_occupancy_f32(int D, bool model, int block, size_t lds);
int uniwin_occupancy_f16(int D, bool model, int block, size_t lds);

int stage_uniwin(const StageArgs &a, bool model) {
    if (a.dtype == HJB_F32) return stage_uniwin_f32(a, model);
    if (a.dtype == HJB_F16S) return stage_uniwin_f16(a, model);
    return 1;                        // float32 arithmetic only
}

int stage_uniwin_occupancy(int dtype, int D, bool model, int block, size_t lds) {
    if (dtype == HJB_F32) return uniwin_occupancy_f32(D, model, block, lds);
    if (dtype == HJB_F16S) return uniwin_occupancy_f16(D, model, block, lds);
    return 0;
}

int stage_uniwin_plan(int D, const DParams *dp, const DNested *dn, int32_t *plan, int n_points, int nA, int nB, int32_t *n_slow) {
    const int grid = (int)((n_points + 255) / 256 < 4096 ? (n_points + 255) / 256 : 4096);
    switch (D) {
        case 4: hipLaunchKernelGGL((k_uniwin_plan<4>), dim3(grid), dim3(256), 0, nullptr, dp, dn, plan, n_points, nA, nB, n_slow); break;
        case 5: hipLaunchKernelGGL((k_uniwin_plan<5>), dim3(grid), dim3(256), 0, nullptr, dp, dn, plan, n_points, nA, nB, n_slow); break;
        case 6: hipLaunchKernelGGL((k_uniwin_plan<6>), dim3(grid), dim3(256), 0, nullptr, dp, dn, plan, n_points, nA, nB, n_slow); break;
        default: return 1;
    }
    return 0;
}

}  // namespace hjb
