#include "stage_colsweep_impl.h"
namespace hjb { int stage_colsweep_c64_f32_g3(const StageArgs &a, int ng, bool dpp) { return colsweep_go_c64<float, 3>(a, ng, dpp); } }
namespace hjb { int stage_colsweep_batch_c64_f32_g3(const StageArgs &a, int n, const DCsBatch *dB, uint32_t mask, int parity, int ng) { return colsweep_go_batch<3, true>(a, n, dB, mask, parity, ng); } }
