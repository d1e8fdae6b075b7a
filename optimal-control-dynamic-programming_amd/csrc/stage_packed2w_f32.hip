#include "stage_packed2_impl.h"
namespace hjb { int stage_packed2w_f32(const StageArgs &a, int mode) { return packed2_go_window<float>(a, mode); } }
