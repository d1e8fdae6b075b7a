#include "stage_colsweep_impl.h"
namespace hjb { int stage_colsweep_c64_f32_g3(const StageArgs &a, int ng, bool dpp) { return colsweep_go_c64<float, 3>(a, ng, dpp); } }
