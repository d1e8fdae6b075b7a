// stage_tile2d.hip - K9 (kernels_tile2d.h): several stages per launch for local 2-D problems
// One translation unit per stage-kernel family (hjbdp_launch.h): built in parallel by __graft_entry__.build().
#include "hjbdp_launch.h"
#include "kernels_tile2d.h"

namespace hjb {

template <typename T, typename TJ>
static int go(const StageArgs &a, const void *plan, int K) {
    const dim3 g(a.grid), b(a.block);
    if (plan)       // stage-invariant per-control data kept in registers
        hipLaunchKernelGGL((k_backup_tile2d_cached<T, TJ>), g, b, 0, a.st, a.dp, (const TilePlan<T> *)plan, (const TJ *)a.Jn, (TJ *)a.Jo, a.idx, K);
    else
        hipLaunchKernelGGL((k_backup_tile2d<T, TJ>), g, b, 0, a.st, a.dp, a.dtb, (const TJ *)a.Jn, (TJ *)a.Jo, a.idx, K);
    return 0;
}

int stage_tile2d(const StageArgs &a, const void *plan, int K) {
    if (a.dtype == HJB_F16S) return go<float, _Float16>(a, plan, K);
    if (a.dtype == HJB_F32) return go<float, float>(a, plan, K);
    return go<double, double>(a, plan, K);
}

int stage_tile2d_plan(int dtype, const DParams *dp, const DTabled *dtb, void *plan, int64_t ne) {
    const dim3 g((unsigned)(ne + 255 < 256 * (int64_t)65536 ? (ne + 255) / 256 : 65536)), b(256);
    if (dtype != HJB_F64) hipLaunchKernelGGL((k_tile2d_plan<float>), g, b, 0, nullptr, dp, dtb, (TilePlan<float> *)plan);
    else hipLaunchKernelGGL((k_tile2d_plan<double>), g, b, 0, nullptr, dp, dtb, (TilePlan<double> *)plan);
    return 0;
}

}  // namespace hjb
