// stage_packed2.hip - variant 4 (K3, kernels_packed2.h): two controls of one state per packed fp32 op; contraction
// modes 0-3.  The two J storage types are compiled in translation units of their own (stage_packed2_f32.hip,
// stage_packed2_f16.hip) so that a cold build runs them in parallel.
#include "hjbdp_launch.h"

namespace hjb {

int stage_packed2_f32(const StageArgs &a, int mode);
int stage_packed2_f16(const StageArgs &a, int mode);
int stage_packed2w_f32(const StageArgs &a, int mode);       // the window modes (2, 3, 5, 6): units of their own, compiled
int stage_packed2w_f16(const StageArgs &a, int mode);       // without the SLP vectoriser (stage_packed2_impl.h)

int stage_packed2(const StageArgs &a, int mode) {
    const bool window = mode == 2 || mode == 3 || mode == 5 || mode == 6;
    if (a.dtype == HJB_F32) return window ? stage_packed2w_f32(a, mode) : stage_packed2_f32(a, mode);
    if (a.dtype == HJB_F16S) return window ? stage_packed2w_f16(a, mode) : stage_packed2_f16(a, mode);
    return 1;                        // float32 arithmetic only
}

}  // namespace hjb
