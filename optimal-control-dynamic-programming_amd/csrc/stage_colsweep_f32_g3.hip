#include "stage_colsweep_impl.h"
namespace hjb { int stage_colsweep_f32_g3(const StageArgs &a, int ng, bool fastcost, bool dpp) { return colsweep_go<float, 3>(a, ng, fastcost, dpp); } }
namespace hjb { int stage_colsweep_batch_f32_g3(const StageArgs &a, int n, const DCsBatch *dB, uint32_t mask, int parity, int ng) { return colsweep_go_batch<3, false>(a, n, dB, mask, parity, ng); } }
