#include "stage_colsweep_impl.h"
namespace hjb { int stage_colsweep_f32_g3(const StageArgs &a, int ng, bool fastcost, bool dpp) { return colsweep_go<float, 3>(a, ng, fastcost, dpp); } }
