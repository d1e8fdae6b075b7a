#include "stage_uniwin_impl.h"
namespace hjb {
int stage_uniwin_f32(const StageArgs &a, bool model) { return uniwin_go<float>(a, model); }
int uniwin_occupancy_f32(int D, bool model, int block, size_t lds) { return uniwin_occupancy_t<float>(D, model, block, lds); }
}
