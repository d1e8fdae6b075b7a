#include "stage_colsweep_impl.h"
namespace hjb { int stage_colsweep_c64_f16_g2(const StageArgs &a, int ng, bool dpp) { return colsweep_go_c64<_Float16, 2>(a, ng, dpp); } }
