// hjbdp_dev.h - device-side problem description shared by all stage kernels.
// gfx950 only.  See DESIGN.md for the data layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hjbdp.h"

namespace hjb {

// one broadcast term on the device: element strides per grid dim (0 = broadcast)
struct DTerm {
    const void *data;
    int32_t stride[HJB_MAX_G];
    int32_t pad;
};

// a state axis: knots, reciprocal spacings, the ordered terms of x_next_a.
// Terms [0, n_prefix) depend on state dims only and are summed once per state;
// terms [n_prefix, n_terms) are evaluated per control (same left-to-right order).
struct DAxis {
    const void *knots;  // [n]   dtype
    const void *rdx;    // [n]   dtype, rdx[i] = 1/(k[i+1]-k[i]), rdx[n-1] = 0
    int32_t n;
    int32_t uniform;    // knots are (numerically) uniform: arithmetic first guess
    double x0, inv_h;   // first-guess map cell ~ (q - x0) * inv_h
    int32_t n_terms, n_prefix;
    DTerm t[HJB_MAX_TERMS];
};

struct DParams {
    int32_t D, C;
    int32_t n[HJB_MAX_D];     // n[D-1] = number of OWNED planes of the last axis
    int32_t m[HJB_MAX_C];
    int64_t n_owned;          // owned states
    int64_t nU;
    int64_t inner;            // states per plane of the last axis
    int64_t jstride[HJB_MAX_D];
    int32_t plane0;           // global plane index of local plane 0 of a J buffer
    int32_t nplanes;          // planes in a J buffer (owned + halo)
    int32_t slab_begin;       // first owned global plane
    int32_t halo_lo;
    int32_t index_base;
    int32_t n_cost, n_cost_prefix;
    int32_t idx_bytes;        // width of a stored argmin label: 4 (int32), 1 (uint8) or 2 (uint16) - hjb_problem.idx_dtype
    int32_t *status;          // device word, set to 1 when a query leaves the slab
    DAxis axis[HJB_MAX_D];
    DTerm cost[HJB_MAX_TERMS];
    // hjb_problem.cost_dtype == HJB_COST_F64: the cost terms in float64 (same strides as cost[]); a stage cost is their ordered
    // sum in double rounded to float32 once (Solver_pos_att.m:800-801).  cost[] then holds float32 copies for host analysis only
    DTerm cost64[HJB_MAX_TERMS];
    int32_t cost_f64;
    int32_t pad_cost;
    // hjb_problem.model (HJB_MODEL_QUAT_EULER321): quaternion tables x4,x5,x6,x7 over (n0,n1,n2), step h
    int32_t model;
    float model_h;
    const void *model_tab[4];
};

// Pointers read out of the parameter structs are generic ("flat") to the compiler.  Everything this library uploads
// lives in global memory; saying so turns flat_load (which also occupies the LDS counter and forces
// s_waitcnt lgkmcnt) into global_load.  Table entries are read as builtin vectors (loadable from any address space).
template <typename T> using gptr = const __attribute__((address_space(1))) T *;
template <typename T> __device__ __forceinline__ gptr<T> as_global(const void *p) { return (gptr<T>)p; }
typedef int i2v __attribute__((ext_vector_type(2)));

// J storage type helpers: J may be stored narrower than the arithmetic type (HJB_F16S: IEEE binary16
// storage, float32 arithmetic; conversion to half rounds to nearest even, widening is exact).
typedef _Float16 half_t;
template <typename T, typename TJ> __device__ __forceinline__ T ldj(const TJ *__restrict__ p, int64_t i) { return (T)p[i]; }
template <typename T, typename TJ> __device__ __forceinline__ void stj(TJ *__restrict__ p, int64_t i, T v) { p[i] = (TJ)v; }

// Argmin labels are stored as int32, uint8 or uint16 (hjb_problem.idx_dtype; MATLAB's U_Optimal_id of
// Solver_pos_att.m:272 holds 9 distinct values).  `bytes` is wave-uniform: a scalar branch around one store.
__device__ __forceinline__ void st_idx(void *__restrict__ base, int64_t i, int32_t v, int32_t bytes) {
    if (bytes == 4) ((int32_t *)base)[i] = v;
    else if (bytes == 1) ((uint8_t *)base)[i] = (uint8_t)v;
    else ((uint16_t *)base)[i] = (uint16_t)v;
}

}  // namespace hjb
