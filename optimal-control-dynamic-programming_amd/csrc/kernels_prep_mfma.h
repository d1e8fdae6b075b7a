// kernels_prep_mfma.h - the affine next-state sum on the matrix cores (BASELINE config 5: "batched-GEMM A.X path on MFMA").
//
// The reference materialises X_next = A(1)*X1 + A(3)*X2 + B(1)*U over the whole state x control grid
// (a_D_M, test/Dynamic_Solver.m:184-188).  libhjbdp never does: x_next is stage-invariant, so the interpolation cell and
// weight of every axis are tabulated once per problem over the axis' own broadcast domain (k_prep_axis_table*), and
// the stage kernels never form A x + B u at all.  The only place a "batched A.X product over all grid states" exists
// is therefore this table build, and this file is its MFMA form, so that rocprof can decide:
//
//   q[r, c] = f[r] + g[c],   f = the ordered sum of all terms but the last (MATLAB's left-to-right order),
//                            g = the last term (the control term B u in every reference solver)
//
// is the rank-2 product [f 1] . [1; g], one v_mfma_f32_32x32x2_f32 per 32 x 32 tile of table entries.  The f32 MFMA is
// bit for bit a k-ordered fmaf chain (MI355X guide): fma(1, g, fma(f, 1, 0)) = round(f + g) - the same single rounding
// as the vector add, so the tables are bit-identical to the term-sum build.  Applies when the last term's grid dims are
// disjoint from the other terms' (float32 arithmetic).  Measured slower than the vector build (profiles/):
// 1024 sums per 64 SIMD-cycles against 64 per ~2.2, and the cell search that follows dominates either way.
#pragma once
#include "hjbdp_dev.h"
#include "kernels_generic.h"

namespace hjb {

struct DPrepSplit {
    int32_t n_row_dims, n_col_dims;
    int32_t row_dim[HJB_MAX_G], col_dim[HJB_MAX_G];       // grid dims (increasing) of the row / column index
    int32_t row_size[HJB_MAX_G], col_size[HJB_MAX_G];
    int32_t row_estride[HJB_MAX_G], col_estride[HJB_MAX_G];   // stride of that dim in the table's entry index
    int32_t n_rows, n_cols;
};

typedef float mfma_acc16 __attribute__((ext_vector_type(16)));

template <int D>
__global__ void __launch_bounds__(256)
k_prep_axis_table_mfma(const DParams *__restrict__ P, int a, DPrepSplit S, int2 *__restrict__ out) {
    __shared__ int s_erow[4][32], s_ecol[4][32];
    const DAxis &ax = P->axis[a];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int tiles_r = (S.n_rows + 31) / 32, tiles_c = (S.n_cols + 31) / 32;
    const int64_t n_tiles = (int64_t)tiles_r * tiles_c;
    const float *kk = static_cast<const float *>(ax.knots);
    const float *rdx = static_cast<const float *>(ax.rdx);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int r0 = (int)(tile % tiles_r) * 32, c0 = (int)(tile / tiles_r) * 32;
        // lanes 0-31: row r0 + l32 (f and the row part of the entry index); lanes 32-63: column c0 + l32 (g, column part)
        int si[D], cj[HJB_MAX_C] = {0, 0, 0};
#pragma unroll
        for (int d = 0; d < D; ++d) si[d] = 0;
        int epart = 0;
        float val = 0.0f;
        if (half == 0) {
            int r = r0 + l32;
            r = r < S.n_rows ? r : S.n_rows - 1;
            for (int i = 0; i < S.n_row_dims; ++i) {
                const int d = S.row_dim[i], id = r % S.row_size[i];
                r /= S.row_size[i];
                epart += id * S.row_estride[i];
                if (d < D) si[d] = id; else cj[d - D] = id;
            }
        } else {
            int c = c0 + l32;
            c = c < S.n_cols ? c : S.n_cols - 1;
            for (int i = 0; i < S.n_col_dims; ++i) {
                const int d = S.col_dim[i], id = c % S.col_size[i];
                c /= S.col_size[i];
                epart += id * S.col_estride[i];
                if (d < D) si[d] = id; else cj[d - D] = id;
            }
        }
        si[D - 1] += P->slab_begin;          // term tables are indexed by GLOBAL grid indices; a domain covers owned planes
        if (half == 0) {
            for (int k = 0; k + 1 < ax.n_terms; ++k) {
                const float x = term_value<float, D>(ax.t[k], si, cj);
                val = (k == 0) ? x : (float)(val + x);
            }
            s_erow[wave][l32] = epart;
        } else {
            val = term_value<float, D>(ax.t[ax.n_terms - 1], si, cj);
            s_ecol[wave][l32] = epart;
        }
        __builtin_amdgcn_wave_barrier();
        // A[row][k]: k = 0 -> f, k = 1 -> 1;   B[k][col]: k = 0 -> 1, k = 1 -> g     (lane l holds k = l >> 5)
        const float a_op = half == 0 ? val : 1.0f;
        const float b_op = half == 0 ? 1.0f : val;
        mfma_acc16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_op, b_op, acc, 0, 0, 0);
        const int ecol = s_ecol[wave][l32];
        const bool col_ok = c0 + l32 < S.n_cols;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;      // C/D layout: col = lane & 31
            if (col_ok && r0 + row < S.n_rows) {
                const float q = acc[reg];
                const int cell = find_cell<float>(kk, ax.n, q, ax.uniform, (float)ax.x0, (float)ax.inv_h);
                const float t = (float)((float)(q - kk[cell]) * rdx[cell]);
                out[s_erow[wave][row] + ecol] = make_int2(cell, __float_as_int(t));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace hjb
