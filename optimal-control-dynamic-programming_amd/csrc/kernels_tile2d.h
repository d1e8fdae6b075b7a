// kernels_tile2d.h - K9: SEVERAL STAGES PER LAUNCH for the 2-D channels (temporal blocking in LDS).
//
// The reference's own workloads - Solver_position.simplified_run (3 x 201x201x3, 5999 stages, :132-141) and
// Solver_attitude.simplified_run (3 x 1000x300x3, :236-247) - are chains of thousands of tiny dependent stages:
// one launch per stage costs ~7 us whatever the kernel does.  Their dynamics move a state by less than one grid
// cell per stage (x+ = x + h v ..., v+ = v + h u / M), so J_k at a state needs J_{k+1} only within +-1 cell.
// A workgroup therefore takes a TX x TY tile PLUS A HALO OF K CELLS, keeps that patch of the cost-to-go in LDS and
// performs K backups on it without leaving the CU: after stage s the patch is valid on the tile grown by K - s
// cells (clipped to the grid - at a grid edge interpolation clamps to the edge cell, which is in the patch).  Halo
// states are recomputed by the neighbouring tiles; every state's arithmetic is the canonical one, so results are
// bit-identical to K single-stage launches.  Applicability (checked once on the host from the axis tables): D = 2,
// whole grid, and for every state and control the interpolation cell of each axis is the state's own cell or the
// one below.  Stage-invariant (cell, t) come from the variant-5 tables; cost terms are evaluated as written.
#pragma once
#include "hjbdp_dev.h"
#include "kernels_generic.h"
#include "kernels_tabled.h"

namespace hjb {

// 16 x 8 owned states and 8 stages per launch (patch 32 x 24 = three states per thread).  Solver_position's channel (201 x 201 x 3),
// 5999 stages alone: 16x16x8 11.63 ms, 8x8x8 11.19, 12x12x8 11.04, 16x8x8 10.32, 16x8x4 10.54, 8x4x4 11.11, 8x8x2 13.77, 16x16x4 12.83,
// 8x8x4 9.82 - but four stages per launch double the launches, and the three channels of simplified_run side by side then take 24.4 ms
// instead of 21 - 22 (the device runs two launch chains at full rate): 16 x 8 x 8 is the form that helps both
// (profiles/r06_xcd_shares_and_spans.log; A/B builds: tools/mkab.sh with -DHJB_TILE_X/Y/K on every unit that includes this header).
#ifndef HJB_TILE_X
#define HJB_TILE_X 16
#endif
#ifndef HJB_TILE_Y
#define HJB_TILE_Y 8
#endif
#ifndef HJB_TILE_K
#define HJB_TILE_K 8
#endif
constexpr int kTileX = HJB_TILE_X, kTileY = HJB_TILE_Y;     // owned states per workgroup
constexpr int kTileK = HJB_TILE_K;                           // stages per launch (halo width)
constexpr int kPatchX = kTileX + 2 * kTileK, kPatchY = kTileY + 2 * kTileK;

// CACHED variant (nU <= kTileMaxU): a thread keeps its <= 4 patch states for the whole launch, so everything that
// does not change from stage to stage - the LDS offset of each control's cell, its two weights and its stage cost -
// is computed once per launch into registers; a stage is then 4 LDS reads + 3 lerps + add + compare per control.
constexpr int kTileMaxU = 4;
constexpr int kTileStatesPerThread = (kPatchX * kPatchY + 255) / 256;

// Everything a backup needs that does not change from stage to stage, per (state, control): built ONCE per problem
// (k_tile2d_plan) so that a launch's prologue is a handful of independent coalesced loads instead of a chain of
// table look-ups and cost-term evaluations per control.
template <typename T> struct TilePlan { int32_t c0, c1; T w0, w1, g; };

template <typename T>
__global__ void __launch_bounds__(256)
k_tile2d_plan(const DParams *__restrict__ P, const DTabled *__restrict__ TB, TilePlan<T> *__restrict__ plan) {
    const int n0 = P->n[0], n1 = P->n[1], nU = (int)P->nU;
    gptr<TabEntry<T>> tab0 = as_global<TabEntry<T>>(TB->ax[0].tab), tab1 = as_global<TabEntry<T>>(TB->ax[1].tab);
    const DTabled::Axis &A0 = TB->ax[0], &A1 = TB->ax[1];
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)n0 * n1 * nU;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int u = (int)(e % nU);
        const int64_t st = e / nU;
        const int gx = (int)(st % n0), gy = (int)(st / n0);
        int si[2] = {gx, gy};
        int cj[HJB_MAX_C] = {u, 0, 0};
        const int o0 = A0.sstride[0] * gx + A0.sstride[1] * gy + A0.cstride[0] * u;
        const int o1 = A1.sstride[0] * gx + A1.sstride[1] * gy + A1.cstride[0] * u;
        TilePlan<T> pl;
        pl.c0 = tab0[o0].cell;
        pl.c1 = tab1[o1].cell;
        pl.w0 = tab0[o0].t;
        pl.w1 = tab1[o1].t;
        T g = (T)0;
        for (int k = 0; k < P->n_cost; ++k) {                    // left to right, like every other kernel
            const T x = term_value<T, 2>(P->cost[k], si, cj);
            g = (k == 0) ? x : (T)(g + x);
        }
        pl.g = g;
        plan[e] = pl;
    }
}

template <typename T, typename TJ>
__global__ void __launch_bounds__(256)
k_backup_tile2d_cached(const DParams *__restrict__ P, const TilePlan<T> *__restrict__ plan_, const TJ *__restrict__ Jn,
                       TJ *__restrict__ Jout, void *__restrict__ idx_out, int K) {
    __shared__ T patch[2][kPatchY * kPatchX];
    const int n0 = P->n[0], n1 = P->n[1];
    const int tiles_x = (n0 + kTileX - 1) / kTileX;
    const int tx0 = (blockIdx.x % tiles_x) * kTileX, ty0 = (blockIdx.x / tiles_x) * kTileY;
    const int px0 = tx0 - K, py0 = ty0 - K;
    const int pw = kTileX + 2 * K, ph = kTileY + 2 * K;
    const int nU = (int)P->nU;                               // <= kTileMaxU, C == 1 (checked on the host)
    gptr<TilePlan<T>> plan = as_global<TilePlan<T>>(plan_);

    int gxs[kTileStatesPerThread], gys[kTileStatesPerThread];     // this thread's patch states (grid coordinates)
    bool live[kTileStatesPerThread];                              // inside the grid
    int qoff[kTileStatesPerThread][kTileMaxU];
    T w0[kTileStatesPerThread][kTileMaxU], w1[kTileStatesPerThread][kTileMaxU], gc[kTileStatesPerThread][kTileMaxU];
#pragma unroll
    for (int m = 0; m < kTileStatesPerThread; ++m) {
        const int i = threadIdx.x + 256 * m;
        const int lx = i % pw, ly = i / pw;
        const int gx = px0 + lx, gy = py0 + ly;
        gxs[m] = gx;
        gys[m] = gy;
        live[m] = i < pw * ph && gx >= 0 && gx < n0 && gy >= 0 && gy < n1;
        if (live[m]) patch[0][ly * kPatchX + lx] = (T)Jn[gx + (int64_t)n0 * gy];
#pragma unroll
        for (int u = 0; u < kTileMaxU; ++u) {
            qoff[m][u] = 0;
            w0[m][u] = w1[m][u] = gc[m][u] = (T)0;
            if (live[m] && u < nU) {
                const int64_t e = ((int64_t)gx + (int64_t)n0 * gy) * nU + u;
                qoff[m][u] = (plan[e].c1 - py0) * kPatchX + (plan[e].c0 - px0);
                w0[m][u] = plan[e].w0;
                w1[m][u] = plan[e].w1;
                gc[m][u] = plan[e].g;
            }
        }
    }
    __syncthreads();

    for (int s = 1; s <= K; ++s) {
        const T *src = patch[(s - 1) & 1];
        T *dst = patch[s & 1];
        const int grow = K - s;
        const int vx0 = tx0 - grow, vx1 = tx0 + kTileX + grow, vy0 = ty0 - grow, vy1 = ty0 + kTileY + grow;
#pragma unroll
        for (int m = 0; m < kTileStatesPerThread; ++m) {
            const int gx = gxs[m], gy = gys[m];
            if (!(live[m] && gx >= vx0 && gx < vx1 && gy >= vy0 && gy < vy1)) continue;
            T best = (T)0;
            int best_u = 0;
#pragma unroll
            for (int u = 0; u < kTileMaxU; ++u) {
                if (u < nU) {
                    const T *q = src + qoff[m][u];
                    const T v00 = q[0], v10 = q[1], v01 = q[kPatchX], v11 = q[kPatchX + 1];
                    const T a = fma_t<T>(w0[m][u], (T)(v10 - v00), v00);
                    const T b = fma_t<T>(w0[m][u], (T)(v11 - v01), v01);
                    const T r = fma_t<T>(w1[m][u], (T)(b - a), a);
                    const T tot = (T)(gc[m][u] + r);
                    if (u == 0 || tot < best) {
                        best = tot;
                        best_u = u;
                    }
                }
            }
            dst[(gy - py0) * kPatchX + (gx - px0)] = (T)(TJ)best;
            if (s == K) {
                Jout[gx + (int64_t)n0 * gy] = (TJ)best;
                if (idx_out) st_idx(idx_out, gx + (int64_t)n0 * gy, best_u + P->index_base, P->idx_bytes);
            }
        }
        __syncthreads();
    }
}

template <typename T, typename TJ>
__global__ void __launch_bounds__(256)
k_backup_tile2d(const DParams *__restrict__ P, const DTabled *__restrict__ TB, const TJ *__restrict__ Jn,
                TJ *__restrict__ Jout, void *__restrict__ idx_out, int K) {
    __shared__ T patch[2][kPatchY * kPatchX];
    const int n0 = P->n[0], n1 = P->n[1];
    const int tiles_x = (n0 + kTileX - 1) / kTileX;
    const int tx0 = (blockIdx.x % tiles_x) * kTileX, ty0 = (blockIdx.x / tiles_x) * kTileY;
    // patch origin in grid coordinates (may be negative: clipped on use)
    const int px0 = tx0 - K, py0 = ty0 - K;
    const int pw = kTileX + 2 * K, ph = kTileY + 2 * K;      // patch extent for this K (<= kPatch*)
    const int nU = (int)P->nU;
    const int C = P->C;
    gptr<TabEntry<T>> tab0 = as_global<TabEntry<T>>(TB->ax[0].tab), tab1 = as_global<TabEntry<T>>(TB->ax[1].tab);
    const DTabled::Axis &A0 = TB->ax[0], &A1 = TB->ax[1];
    const int m1 = P->m[1], m2 = P->m[2];

    // ---- load the patch of J_{k+1} (grid part only) ----------------------------------------------
    for (int i = threadIdx.x; i < pw * ph; i += blockDim.x) {
        const int lx = i % pw, ly = i / pw;
        const int gx = px0 + lx, gy = py0 + ly;
        if (gx >= 0 && gx < n0 && gy >= 0 && gy < n1) patch[0][ly * kPatchX + lx] = (T)Jn[gx + (int64_t)n0 * gy];
    }
    __syncthreads();

    for (int s = 1; s <= K; ++s) {
        const T *src = patch[(s - 1) & 1];
        T *dst = patch[s & 1];
        const int grow = K - s;                              // the tile grown by this many cells is valid after stage s
        const int vx0 = max(tx0 - grow, 0), vx1 = min(tx0 + kTileX + grow, n0);
        const int vy0 = max(ty0 - grow, 0), vy1 = min(ty0 + kTileY + grow, n1);
        const int vw = vx1 - vx0, vh = vy1 - vy0;
        for (int i = threadIdx.x; i < vw * vh; i += blockDim.x) {
            const int gx = vx0 + i % vw, gy = vy0 + i / vw;
            int si[2] = {gx, gy};
            const int off0 = A0.sstride[0] * gx + A0.sstride[1] * gy;
            const int off1 = A1.sstride[0] * gx + A1.sstride[1] * gy;
            int cj[HJB_MAX_C] = {0, 0, 0};
            T gpre = (T)0;
            for (int k = 0; k < P->n_cost_prefix; ++k) {
                const T x = term_value<T, 2>(P->cost[k], si, cj);
                gpre = (k == 0) ? x : (T)(gpre + x);
            }
            T best = (T)0;
            int best_u = 0;
            for (int u = 0; u < nU; ++u) {
                const int o0 = off0 + A0.cstride[0] * cj[0] + A0.cstride[1] * cj[1] + A0.cstride[2] * cj[2];
                const int o1 = off1 + A1.cstride[0] * cj[0] + A1.cstride[1] * cj[1] + A1.cstride[2] * cj[2];
                const int c0 = tab0[o0].cell, c1 = tab1[o1].cell;
                const T t0 = tab0[o0].t, t1 = tab1[o1].t;
                const T *q = src + (c1 - py0) * kPatchX + (c0 - px0);
                const T v00 = q[0], v10 = q[1], v01 = q[kPatchX], v11 = q[kPatchX + 1];
                const T a = fma_t<T>(t0, (T)(v10 - v00), v00);
                const T b = fma_t<T>(t0, (T)(v11 - v01), v01);
                const T r = fma_t<T>(t1, (T)(b - a), a);
                T g = gpre;
                for (int k = P->n_cost_prefix; k < P->n_cost; ++k) {
                    const T x = term_value<T, 2>(P->cost[k], si, cj);
                    g = (k == 0) ? x : (T)(g + x);
                }
                const T tot = (T)(g + r);
                if (u == 0 || tot < best) {
                    best = tot;
                    best_u = u;
                }
                if (C == 1) {
                    ++cj[0];
                } else if (C == 2) {
                    if (++cj[1] == m1) { cj[1] = 0; ++cj[0]; }
                } else {
                    if (++cj[2] == m2) {
                        cj[2] = 0;
                        if (++cj[1] == m1) { cj[1] = 0; ++cj[0]; }
                    }
                }
            }
            const T stored = (T)(TJ)best;                    // what a one-stage launch would have left in memory
            dst[(gy - py0) * kPatchX + (gx - px0)] = stored;
            if (s == K) {                                    // here the valid region is the tile itself
                Jout[gx + (int64_t)n0 * gy] = (TJ)best;
                if (idx_out) {
                    int64_t label;
                    if (C == 1) {
                        label = best_u;
                    } else if (C == 2) {
                        const int j1 = best_u % m1, j0 = best_u / m1;
                        label = j0 + (int64_t)P->m[0] * j1;
                    } else {
                        const int j2 = best_u % m2;
                        const int rr = best_u / m2;
                        const int j1 = rr % m1, j0 = rr / m1;
                        label = j0 + (int64_t)P->m[0] * (j1 + (int64_t)m1 * j2);
                    }
                    st_idx(idx_out, gx + (int64_t)n0 * gy, (int32_t)(label + P->index_base), P->idx_bytes);
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace hjb
