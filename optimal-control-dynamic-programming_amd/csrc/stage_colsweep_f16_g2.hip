#include "stage_colsweep_impl.h"
namespace hjb { int stage_colsweep_f16_g2(const StageArgs &a, int ng, bool fastcost, bool dpp) { return colsweep_go<_Float16, 2>(a, ng, fastcost, dpp); } }
