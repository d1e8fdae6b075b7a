// stage_colsweep.hip - variant 7 (K10, kernels_colsweep.h): the pos-att column sweep.  96 instantiations (group
// axis x groups per plan x cost form x one-load form x J storage): compiled as four translation units
// (stage_colsweep_{f32,f16}_g{2,3}.hip) so that a cold build runs them in parallel; the 48 of cost form 2 (float64 cost
// terms, hjb_problem.cost_dtype) in four more (stage_colsweep_c64_*.hip).
#include "hjbdp_launch.h"

namespace hjb {

int stage_colsweep_f32_g2(const StageArgs &a, int ng, bool fastcost, bool dpp);
int stage_colsweep_f32_g3(const StageArgs &a, int ng, bool fastcost, bool dpp);
int stage_colsweep_f16_g2(const StageArgs &a, int ng, bool fastcost, bool dpp);
int stage_colsweep_f16_g3(const StageArgs &a, int ng, bool fastcost, bool dpp);

int stage_colsweep_c64_f32_g2(const StageArgs &a, int ng, bool dpp);
int stage_colsweep_c64_f32_g3(const StageArgs &a, int ng, bool dpp);
int stage_colsweep_c64_f16_g2(const StageArgs &a, int ng, bool dpp);
int stage_colsweep_c64_f16_g3(const StageArgs &a, int ng, bool dpp);

int stage_colsweep_batch_f32_g2(const StageArgs &a, int n, const DCsBatch *dB, uint32_t mask, int parity, int ng);
int stage_colsweep_batch_f32_g3(const StageArgs &a, int n, const DCsBatch *dB, uint32_t mask, int parity, int ng);
int stage_colsweep_batch_c64_f32_g2(const StageArgs &a, int n, const DCsBatch *dB, uint32_t mask, int parity, int ng);
int stage_colsweep_batch_c64_f32_g3(const StageArgs &a, int n, const DCsBatch *dB, uint32_t mask, int parity, int ng);

// n problems of one shape in one launch (float32 J, the usual cost shape, the one-load form): a.grid = the largest workgroup count
int stage_colsweep_batch(const StageArgs &a, int n, const DCsBatch *dB, uint32_t mask, int parity, int gax, int ng, bool c64) {
    if (a.dtype != HJB_F32) return 1;
    if (c64) return gax == 3 ? stage_colsweep_batch_c64_f32_g3(a, n, dB, mask, parity, ng) : stage_colsweep_batch_c64_f32_g2(a, n, dB, mask, parity, ng);
    return gax == 3 ? stage_colsweep_batch_f32_g3(a, n, dB, mask, parity, ng) : stage_colsweep_batch_f32_g2(a, n, dB, mask, parity, ng);
}

// costform: 0 general control terms, 1 state terms + one control term (float32), 2 the same summed in float64
int stage_colsweep(const StageArgs &a, int gax, int ng, int costform, bool dpp) {
    const bool fastcost = costform != 0;
    if (costform == 2) {
        if (a.dtype == HJB_F32) return gax == 3 ? stage_colsweep_c64_f32_g3(a, ng, dpp) : stage_colsweep_c64_f32_g2(a, ng, dpp);
        if (a.dtype == HJB_F16S) return gax == 3 ? stage_colsweep_c64_f16_g3(a, ng, dpp) : stage_colsweep_c64_f16_g2(a, ng, dpp);
        return 1;
    }
    if (a.dtype == HJB_F32) return gax == 3 ? stage_colsweep_f32_g3(a, ng, fastcost, dpp) : stage_colsweep_f32_g2(a, ng, fastcost, dpp);
    if (a.dtype == HJB_F16S) return gax == 3 ? stage_colsweep_f16_g3(a, ng, fastcost, dpp) : stage_colsweep_f16_g2(a, ng, fastcost, dpp);
    return 1;                        // float32 arithmetic only
}

}  // namespace hjb
