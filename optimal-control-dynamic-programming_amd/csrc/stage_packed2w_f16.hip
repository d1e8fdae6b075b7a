#include "stage_packed2_impl.h"
namespace hjb { int stage_packed2w_f16(const StageArgs &a, int mode) { return packed2_go_window<_Float16>(a, mode); } }
