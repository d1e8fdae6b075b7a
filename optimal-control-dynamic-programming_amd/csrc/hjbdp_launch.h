// hjbdp_launch.h - the seam between the host side of libhjbdp (hjbdp_setup.hip: launch_stage) and the stage-kernel families, each
// compiled in a translation unit of its own (stage_*.hip) so that a cold build runs them in parallel
// (__graft_entry__.build()).  Plain arguments only: a family knows nothing of the handle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hjbdp_dev.h"

namespace hjb {

struct DNested;
struct DTabled;
struct DColSweep;
struct DUniwin;
struct DCsBatch;

struct StageArgs {
    unsigned grid = 1, block = 256;
    size_t lds = 0;
    hipStream_t st = nullptr;
    bool idx32 = false;            // variant 5: every index fits 31 bits (the 32-bit form of the table kernel)
    int dtype = HJB_F32;           // HJB_F32 / HJB_F64 / HJB_F16S (float32 arithmetic, binary16 J storage)
    int D = 1;
    const DParams *dp = nullptr;
    const DNested *dn = nullptr;
    const DTabled *dtb = nullptr;
    const DColSweep *dcs = nullptr;
    const DUniwin *duw = nullptr;  // variant 4 modes 7 / 8 (kernels_uniwin.h)
    const void *Jn = nullptr;
    void *Jo = nullptr;
    void *idx = nullptr;           // argmin labels, width DParams::idx_bytes
};

// Each returns 0 when a kernel was enqueued, 1 when the family has no instantiation for (dtype, D, flags).
int stage_generic(const StageArgs &a);                                           // variant 0
int stage_nested(const StageArgs &a, bool fast);                                 // variant 1
int stage_packed(const StageArgs &a);                                            // variant 2
int stage_ctrlsplit(const StageArgs &a, bool j_in_lds);                          // variant 3
int stage_packed2(const StageArgs &a, int mode);                                 // variant 4
int stage_uniwin(const StageArgs &a, bool model);                                // variant 4, modes 7 / 8 (K15)
int stage_uniwin_occupancy(int dtype, int D, bool model, int block, size_t lds); // workgroups of `block` threads one CU holds
int stage_uniwin_plan(int D, const DParams *dp, const DNested *dn, int32_t *plan, int n_points, int nA, int nB, int32_t *n_slow);
int stage_tabled(const StageArgs &a);                                            // variant 5
int stage_tabled_batch(const StageArgs &a, int n, const DCsBatch &hB, uint32_t mask, int parity);   // variant 5 (32-bit form), n problems in one launch (record by value)
int stage_rowwise(const StageArgs &a, bool lean);                                // variant 6
int stage_colsweep(const StageArgs &a, int gax, int ng, int costform, bool dpp);    // variant 7 (costform: 0 general, 1 fast, 2 fast in float64)
int stage_colcoop(const StageArgs &a, int gax, int ng, bool fastcost);           // variant 7, cooperative form
int stage_colsweep_batch(const StageArgs &a, int n, const DCsBatch *dB, uint32_t mask, int parity, int gax, int ng, bool c64);   // variant 7, n problems in one launch
int stage_tile2d(const StageArgs &a, const void *plan, int K);                   // K9 (several stages per launch)
int stage_tile2d_plan(int dtype, const DParams *dp, const DTabled *dtb, void *plan, int64_t n_entries);

}  // namespace hjb
