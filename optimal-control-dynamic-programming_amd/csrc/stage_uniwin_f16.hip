#include "stage_uniwin_impl.h"
namespace hjb {
int stage_uniwin_f16(const StageArgs &a, bool model) { return uniwin_go<_Float16>(a, model); }
int uniwin_occupancy_f16(int D, bool model, int block, size_t lds) { return uniwin_occupancy_t<_Float16>(D, model, block, lds); }
}
