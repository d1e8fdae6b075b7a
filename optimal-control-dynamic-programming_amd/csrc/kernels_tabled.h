// kernels_tabled.h - variant 5: the general stage kernel on precomputed tables.
//
// Every axis' interpolation cell and weight is stage-invariant, so hjb_create
// precomputes them once (k_prep_axis_table, canonical arithmetic) over each axis'
// own broadcast domain - e.g. pos-att (Solver_pos_att.m:299-328): x+ over (x,v),
// v+ over (v,u), theta+ over (theta,w), w+ over (w,u): four tiny tables.  The stage
// kernel then does, per control, one 8/16-byte lookup for each axis that depends on
// a control (the others are looked up once per state), the 2^D-corner gather, the
// lerps, the cost terms and the strict-< argmin: no term sums, no searches.
// Any D <= 6, C <= 3, float32/float64, slabs.  Bit-identical to variants 0-4.
#pragma once
#include "hjbdp_dev.h"
#include "kernels_generic.h"

namespace hjb {

template <typename T> struct TabEntry { int32_t cell; T t; };   // float: 8 B, double: 16 B

// The two axis-0 neighbours of a corner in ONE load: axis 0 is contiguous in every J layout of this library (DParams::jstride[0] == 1),
// so corners 2c and 2c + 1 sit side by side - an 8-byte (float32), 4-byte (binary16) or 16-byte (float64) load at element alignment.
typedef float tab_f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef _Float16 tab_h2u __attribute__((ext_vector_type(2), aligned(2)));
typedef double tab_d2u __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void tab_load_pair(const float *__restrict__ Jn, int64_t off, float &a, float &b) {
    const tab_f2u p = *reinterpret_cast<const tab_f2u *>(Jn + off);
    a = p.x; b = p.y;
}
__device__ __forceinline__ void tab_load_pair(const _Float16 *__restrict__ Jn, int64_t off, float &a, float &b) {
    const tab_h2u p = *reinterpret_cast<const tab_h2u *>(Jn + off);
    a = (float)p.x; b = (float)p.y;
}
__device__ __forceinline__ void tab_load_pair(const double *__restrict__ Jn, int64_t off, double &a, double &b) {
    const tab_d2u p = *reinterpret_cast<const tab_d2u *>(Jn + off);
    a = p.x; b = p.y;
}

// Workgroup b runs on XCD b % 8 (each XCD has its own L2).  With "workgroup b takes states [256 b, 256 b + 256)" every XCD walks the
// whole grid and its L2 holds all of J; here XCD x takes the x-th CONTIGUOUS share of every grid-sized span of workgroups instead
// (its workgroups b = x, x + 8, ...: G / 8 of them, one more for x < G % 8), so an L2 holds one region of J and its halo.  The host
// sizes the launch so that the spans are equally long (choose_launch): a short last span would fall to the first XCDs alone.
__device__ __forceinline__ unsigned xcd_share(unsigned b, unsigned G) {
    const unsigned x = b & 7u, q = G >> 3, r = G & 7u;
    return x * q + (x < r ? x : r) + (b >> 3);
}

struct DTabled {
    struct Axis {
        const void *tab;
        int32_t sstride[HJB_MAX_D];   // entry strides along the state dims in the domain (0 otherwise)
        int32_t cstride[HJB_MAX_C];   // entry strides along the control dims in the domain (0 otherwise)
        int32_t has_ctrl;             // domain contains a control dim -> looked up per control
        int32_t pad;
    } ax[HJB_MAX_D];
};

// one thread per domain entry (see k_prep_axis_table); T-typed weight
template <typename T, int D>
__global__ void __launch_bounds__(256)
k_prep_axis_table_t(const DParams *__restrict__ P, int a, const int32_t *__restrict__ dom_size, int64_t n_entries,
                    TabEntry<T> *__restrict__ out) {
    const DAxis &ax = P->axis[a];
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_entries;
         e += (int64_t)gridDim.x * blockDim.x) {
        int si[D];
        int cj[HJB_MAX_C] = {0, 0, 0};
        int64_t r = e;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int sz = dom_size[d];
            si[d] = (int)(r % sz);
            r /= sz;
        }
#pragma unroll
        for (int c = 0; c < HJB_MAX_C; ++c) {
            const int sz = dom_size[D + c];
            cj[c] = (int)(r % sz);
            r /= sz;
        }
        si[D - 1] += P->slab_begin;
        T q = (T)0;
        for (int k = 0; k < ax.n_terms; ++k) {
            T x = term_value<T, D>(ax.t[k], si, cj);
            q = (k == 0) ? x : (T)(q + x);
        }
        const T *kk = static_cast<const T *>(ax.knots);
        const int cell = find_cell<T>(kk, ax.n, q, ax.uniform, (T)ax.x0, (T)ax.inv_h);
        TabEntry<T> ent;
        ent.cell = cell;
        ent.t = (T)((T)(q - kk[cell]) * static_cast<const T *>(ax.rdx)[cell]);
        out[e] = ent;
    }
}

template <typename T, typename TJ, int D>
__global__ void __launch_bounds__(256)
k_backup_tabled(const DParams *__restrict__ P, const DTabled *__restrict__ TB, const TJ *__restrict__ Jn,
                TJ *__restrict__ Jout, void *__restrict__ idx_out) {
    const int C = P->C;
    const int64_t n_owned = P->n_owned;
    const int64_t nU = P->nU;
    const int plane0 = P->plane0, nplanes = P->nplanes;
    for (int64_t ls = xcd_share(blockIdx.x, gridDim.x) * (int64_t)blockDim.x + threadIdx.x; ls < n_owned;
         ls += (int64_t)gridDim.x * blockDim.x) {
        int si[D];          // global indices (cost term tables)
        int sl[D];          // local index along the last axis (axis tables cover owned planes)
        {
            int64_t r = ls;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                int na = P->n[a];
                si[a] = (int)(r % na);
                sl[a] = si[a];
                r /= na;
            }
            si[D - 1] += P->slab_begin;
        }
        int cj[HJB_MAX_C] = {0, 0, 0};
        int64_t aoff[D];
        int cell[D];
        T tw[D];
        bool bad0 = false;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const DTabled::Axis &A = TB->ax[a];
            int64_t off = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) off += (int64_t)A.sstride[d] * sl[d];
            aoff[a] = off;
            if (!A.has_ctrl) {
                const TabEntry<T> e = static_cast<const TabEntry<T> *>(A.tab)[off];
                cell[a] = e.cell;
                tw[a] = e.t;
                if (a == D - 1) {
                    cell[a] -= plane0;
                    if (cell[a] < 0 || cell[a] + 1 >= nplanes) { bad0 = true; cell[a] = cell[a] < 0 ? 0 : nplanes - 2; }
                }
            }
        }
        if (bad0) *P->status = 1;
        T gpre = (T)0;
        double gpre64 = 0.0;                     // cost_dtype F64: the state part of the cost in double (Solver_pos_att.m:800)
        const bool c64 = P->cost_f64 != 0;
        if (c64) {
            for (int k = 0; k < P->n_cost_prefix; ++k) {
                const double x = term_value<double, D>(P->cost64[k], si, cj);
                gpre64 = (k == 0) ? x : gpre64 + x;
            }
        } else {
            for (int k = 0; k < P->n_cost_prefix; ++k) {
                T x = term_value<T, D>(P->cost[k], si, cj);
                gpre = (k == 0) ? x : (T)(gpre + x);
            }
        }
        T best = (T)0;
        int64_t best_u = 0;
        for (int64_t u = 0; u < nU; ++u) {
            int64_t base = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const DTabled::Axis &A = TB->ax[a];
                if (A.has_ctrl) {
                    int64_t off = aoff[a];
#pragma unroll
                    for (int c = 0; c < HJB_MAX_C; ++c) off += (int64_t)A.cstride[c] * cj[c];
                    const TabEntry<T> e = static_cast<const TabEntry<T> *>(A.tab)[off];
                    int cl = e.cell;
                    tw[a] = e.t;
                    if (a == D - 1) {
                        cl -= plane0;
                        if (cl < 0 || cl + 1 >= nplanes) { *P->status = 1; cl = cl < 0 ? 0 : nplanes - 2; }
                    }
                    cell[a] = cl;
                }
                base += P->jstride[a] * cell[a];
            }
            T v[1 << D];
#pragma unroll
            for (int c = 0; c < (1 << D); ++c) {
                int64_t off = base;
#pragma unroll
                for (int a = 0; a < D; ++a)
                    if (c & (1 << a)) off += P->jstride[a];
                v[c] = ldj<T, TJ>(Jn, off);
            }
#pragma unroll
            for (int a = 0; a < D; ++a) {
#pragma unroll
                for (int j = 0; j < (1 << (D - 1 - a)); ++j)
                    v[j] = fma_t<T>(tw[a], (T)(v[2 * j + 1] - v[2 * j]), v[2 * j]);
            }
            T g = gpre;
            if (c64) {                           // the control part added in double, ONE rounding to the arithmetic type
                double g64 = gpre64;
                for (int k = P->n_cost_prefix; k < P->n_cost; ++k) {
                    const double x = term_value<double, D>(P->cost64[k], si, cj);
                    g64 = (k == 0) ? x : g64 + x;
                }
                g = (T)g64;
            } else {
                for (int k = P->n_cost_prefix; k < P->n_cost; ++k) {
                    T x = term_value<T, D>(P->cost[k], si, cj);
                    g = (k == 0) ? x : (T)(g + x);
                }
            }
            const T tot = (T)(g + v[0]);
            if (u == 0 || tot < best) {
                best = tot;
                best_u = u;
            }
            if (C == 1) {
                ++cj[0];
            } else if (C == 2) {
                if (++cj[1] == P->m[1]) { cj[1] = 0; ++cj[0]; }
            } else {
                if (++cj[2] == P->m[2]) {
                    cj[2] = 0;
                    if (++cj[1] == P->m[1]) { cj[1] = 0; ++cj[0]; }
                }
            }
        }
        int64_t label;
        if (C == 1) {
            label = best_u;
        } else if (C == 2) {
            int64_t j1 = best_u % P->m[1], j0 = best_u / P->m[1];
            label = j0 + (int64_t)P->m[0] * j1;
        } else {
            int64_t j2 = best_u % P->m[2];
            int64_t rr = best_u / P->m[2];
            int64_t j1 = rr % P->m[1], j0 = rr / P->m[1];
            label = j0 + (int64_t)P->m[0] * (j1 + (int64_t)P->m[1] * j2);
        }
        const int64_t in_plane = ls % P->inner, pl = ls / P->inner;
        stj<T, TJ>(Jout, in_plane + P->inner * (pl + P->halo_lo), best);
        if (idx_out) st_idx(idx_out, ls, (int32_t)(label + P->index_base), P->idx_bytes);
    }
}


// ---- the same kernel in 32-bit index arithmetic (round 5) ----------------------------------------------------------------------
// k_backup_tabled above is written for any size: 64-bit state indices, offsets and strides throughout.  On gfx950 that is what
// it spends its time on - per control ~130 vector and ~200 SCALAR instructions of 64-bit multiplies, carries and generic term
// loops around 16 loads and 34 floating-point operations (the reference's pos-att grid: 21.6 us per stage, the same
// backups per second on a grid sixteen times larger).  When every index
// fits 31 bits (hjb_create checks: owned states, the haloed J, every axis table, states x controls for the cost tables) this
// form runs instead:
//   * 32-bit indices; the state index taken apart by 32-bit unsigned division;
//   * the corner offsets of a cell - sums of the J strides of axes 1 .. D-1 - formed ONCE per kernel (wave-uniform), a
//     corner pair's address is then one vector add; the axis-0 pair of a corner is ONE load (axis 0 is contiguous);
//   * table entries and cost terms through global, not flat, loads; a control-dependent cost term's offset is its state
//     part (once per state) plus a wave-uniform control part (scalar unit);
//   * the control dims' counters stay scalar (one running set per kernel loop trip);
//   * the control-dependent axes' table entries are read ONE CONTROL AHEAD (requested behind this control's corners, before anything
//     waits): one flight per control instead of entries -> corners in line.
// Each of these alone moved nothing (the kernel is co-limited by its integer work and its corner loads: profiles/r05_k7_pmc.json);
// together: the reference's pos-att grid 21.65 -> 20.55 us per stage, 60x60x40x30 0.2228 -> 0.2003 ms (profiles/r05_k7_batch_experiment.log).
// The floating-point operations, their operands and their order are the generic kernel's: same bits.
constexpr int kTab32MaxCt = 4;      // control-dependent cost terms whose state offsets are hoisted (more: generic term evaluation)
// The 32-bit form advances its state index as a signed int by the launch's thread count: eligibility (hjbdp_setup.hip) keeps every
// index below kTab32Lim and the launch (launch_stage) runs the 64-bit form when the launch has more than kTab32MaxThreads threads -
// the two facts the loop counter's range rests on, tied together here
constexpr int64_t kTab32MaxThreads = (int64_t)1 << 26;
constexpr int64_t kTab32Lim = ((int64_t)1 << 31) - kTab32MaxThreads;
static_assert(kTab32Lim - 1 + kTab32MaxThreads <= (int64_t)INT32_MAX, "k_backup_tabled32: the last index plus one grid stride must fit a signed int");

template <typename T, int D>
__device__ __forceinline__ int tab32_state_off(const DTerm &t, const int (&si)[D]) {
    int off = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) off += t.stride[a] * si[a];
    return off;
}
__device__ __forceinline__ int tab32_ctrl_off(const int32_t *stride_c, const int (&cj)[HJB_MAX_C]) {      // wave-uniform
    int off = 0;
#pragma unroll
    for (int c = 0; c < HJB_MAX_C; ++c) off += stride_c[c] * cj[c];
    return off;
}

// The kernel's body: the states of workgroup `bx` of `gx` (the launch's own blockIdx.x / gridDim.x in k_backup_tabled32; the
// problem's share of a batched launch in k_backup_tabled32_batch).
template <typename T, typename TJ, int D>
__device__ __forceinline__ void tabled32_body(const DParams *__restrict__ P, const DTabled *__restrict__ TB, const TJ *__restrict__ Jn,
                                              TJ *__restrict__ Jout, void *__restrict__ idx_out, unsigned bx, unsigned gx) {
    const int C = P->C;
    const int n_owned = (int)P->n_owned;
    const int nU = (int)P->nU;
    const int plane0 = P->plane0, nplanes = P->nplanes;
    const int npre = P->n_cost_prefix, ncost = P->n_cost;
    const int nct = ncost - npre;                                   // control-dependent cost terms
    const bool c64 = P->cost_f64 != 0;
    const bool hoist = nct <= kTab32MaxCt;
    int js[D];
#pragma unroll
    for (int a = 0; a < D; ++a) js[a] = a == 0 ? 1 : (int)P->jstride[a];
    int poff[1 << (D > 1 ? D - 1 : 0)];                             // element offset of corner pair p inside a cell (axes 1 .. D-1)
#pragma unroll
    for (int p = 0; p < (1 << (D > 1 ? D - 1 : 0)); ++p) {
        int o = 0;
#pragma unroll
        for (int a = 1; a < D; ++a)
            if (p & (1 << (a - 1))) o += js[a];
        poff[p] = o;
    }
    const int m1 = P->m[1], m2 = P->m[2];
    for (int ls = (int)(xcd_share(bx, gx) * blockDim.x + threadIdx.x); ls < n_owned; ls += (int)(gx * blockDim.x)) {
        int si[D], sl[D];
        {
            uint32_t r = (uint32_t)ls;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const uint32_t na = (uint32_t)P->n[a];
                const uint32_t q = r / na;
                si[a] = (int)(r - q * na);
                sl[a] = si[a];
                r = q;
            }
            si[D - 1] += P->slab_begin;
        }
        int aoff[D], cell[D];
        T tw[D];
        bool bad0 = false;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const DTabled::Axis &A = TB->ax[a];
            int off = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) off += A.sstride[d] * sl[d];
            aoff[a] = off;
            if (!A.has_ctrl) {
                cell[a] = as_global<TabEntry<T>>(A.tab)[off].cell;
                tw[a] = as_global<TabEntry<T>>(A.tab)[off].t;
                if (a == D - 1) {
                    cell[a] -= plane0;
                    if (cell[a] < 0 || cell[a] + 1 >= nplanes) { bad0 = true; cell[a] = cell[a] < 0 ? 0 : nplanes - 2; }
                }
            }
        }
        if (bad0) *P->status = 1;
        int cj[HJB_MAX_C] = {0, 0, 0};
        T gpre = (T)0;
        double gpre64 = 0.0;
        if (c64) {
            for (int k = 0; k < npre; ++k) {
                const double x = as_global<double>(P->cost64[k].data)[tab32_state_off<double, D>(P->cost64[k], si)];
                gpre64 = (k == 0) ? x : gpre64 + x;
            }
        } else {
            for (int k = 0; k < npre; ++k) {
                const T x = as_global<T>(P->cost[k].data)[tab32_state_off<T, D>(P->cost[k], si)];
                gpre = (k == 0) ? x : (T)(gpre + x);
            }
        }
        int ct_off[kTab32MaxCt];                                    // state part of the control-dependent cost terms' offsets
#pragma unroll
        for (int k = 0; k < kTab32MaxCt; ++k)
            ct_off[k] = (hoist && k < nct) ? tab32_state_off<T, D>(c64 ? P->cost64[npre + k] : P->cost[npre + k], si) : 0;
        T best = (T)0;
        int best_u = 0;
        int nx_cell[D];
        T nx_t[D];
        auto fetch32 = [&](const int (&cq)[HJB_MAX_C]) {
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const DTabled::Axis &A = TB->ax[a];
                if (A.has_ctrl) {
                    const int off = aoff[a] + tab32_ctrl_off(A.cstride, cq);
                    nx_cell[a] = as_global<TabEntry<T>>(A.tab)[off].cell;
                    nx_t[a] = as_global<TabEntry<T>>(A.tab)[off].t;
                }
            }
        };
        fetch32(cj);
        for (int u = 0; u < nU; ++u) {
            int base = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const DTabled::Axis &A = TB->ax[a];
                if (A.has_ctrl) {
                    int cl = nx_cell[a];
                    tw[a] = nx_t[a];
                    if (a == D - 1) {
                        cl -= plane0;
                        if (cl < 0 || cl + 1 >= nplanes) { *P->status = 1; cl = cl < 0 ? 0 : nplanes - 2; }
                    }
                    cell[a] = cl;
                }
                base += js[a] * cell[a];
            }
            T v[1 << D];
#pragma unroll
            for (int p = 0; p < (1 << (D > 1 ? D - 1 : 0)); ++p) tab_load_pair(Jn, (int64_t)(base + poff[p]), v[2 * p], v[2 * p + 1]);
            int cjn[HJB_MAX_C] = {cj[0], cj[1], cj[2]};
            if (C == 1) {
                ++cjn[0];
            } else if (C == 2) {
                if (++cjn[1] == m1) { cjn[1] = 0; ++cjn[0]; }
            } else {
                if (++cjn[2] == m2) {
                    cjn[2] = 0;
                    if (++cjn[1] == m1) { cjn[1] = 0; ++cjn[0]; }
                }
            }
            if (u + 1 < nU) fetch32(cjn);
            T g = gpre;
            if (c64) {                           // the control part added in double, ONE rounding to the arithmetic type
                double g64 = gpre64;
                if (hoist) {
#pragma unroll
                    for (int k = 0; k < kTab32MaxCt; ++k)
                        if (k < nct) {
                            const DTerm &t = P->cost64[npre + k];
                            const double x = as_global<double>(t.data)[ct_off[k] + tab32_ctrl_off(t.stride + D, cj)];
                            g64 = (npre + k == 0) ? x : g64 + x;
                        }
                } else {
                    for (int k = npre; k < ncost; ++k) {
                        const double x = term_value<double, D>(P->cost64[k], si, cj);
                        g64 = (k == 0) ? x : g64 + x;
                    }
                }
                g = (T)g64;
            } else if (hoist) {
#pragma unroll
                for (int k = 0; k < kTab32MaxCt; ++k)
                    if (k < nct) {
                        const DTerm &t = P->cost[npre + k];
                        const T x = as_global<T>(t.data)[ct_off[k] + tab32_ctrl_off(t.stride + D, cj)];
                        g = (npre + k == 0) ? x : (T)(g + x);
                    }
            } else {
                for (int k = npre; k < ncost; ++k) {
                    const T x = term_value<T, D>(P->cost[k], si, cj);
                    g = (k == 0) ? x : (T)(g + x);
                }
            }
#pragma unroll
            for (int a = 0; a < D; ++a) {
#pragma unroll
                for (int j = 0; j < (1 << (D - 1 - a)); ++j)
                    v[j] = fma_t<T>(tw[a], (T)(v[2 * j + 1] - v[2 * j]), v[2 * j]);
            }
            const T tot = (T)(g + v[0]);
            if (u == 0 || tot < best) {
                best = tot;
                best_u = u;
            }
            cj[0] = cjn[0]; cj[1] = cjn[1]; cj[2] = cjn[2];
        }
        int label;
        if (C == 1) {
            label = best_u;
        } else if (C == 2) {
            const int j1 = best_u % m1, j0 = best_u / m1;
            label = j0 + P->m[0] * j1;
        } else {
            const int j2 = best_u % m2;
            const int rr = best_u / m2;
            const int j1 = rr % m1, j0 = rr / m1;
            label = j0 + P->m[0] * (j1 + m1 * j2);
        }
        const uint32_t inner = (uint32_t)P->inner;
        const uint32_t pl = (uint32_t)ls / inner, in_plane = (uint32_t)ls - pl * inner;
        stj<T, TJ>(Jout, (int64_t)(in_plane + inner * (pl + (uint32_t)P->halo_lo)), best);
        if (idx_out) st_idx(idx_out, ls, (int32_t)(label + P->index_base), P->idx_bytes);
    }
}

template <typename T, typename TJ, int D>
__global__ void __launch_bounds__(256)
k_backup_tabled32(const DParams *__restrict__ P, const DTabled *__restrict__ TB, const TJ *__restrict__ Jn,
                  TJ *__restrict__ Jout, void *__restrict__ idx_out) {
    tabled32_body<T, TJ, D>(P, TB, Jn, Jout, idx_out, blockIdx.x, gridDim.x);
}

// ---- several problems, one launch (hjb_solve_batch) -------------------------------------------------------------------------------
// The reference's simplified_run sweeps its channels one after the other (attitude-control/Solver_attitude.m:196-259: three channels of
// 3e5 states x 5999 stages; pos-att/Solver_pos_att.m:197-242: four of 2.7e5 x 1999): a stage kernel of one of them is a launch boundary
// plus one wave's chain of round trips, and of several such chains on streams of their own the device runs two at full rate.  Here
// blockIdx.y = the problem; each keeps its own parameters, tables, buffers and workgroup count; `mask` drops the problems whose
// monitor has stopped them; `parity` says which of a problem's two J buffers is the input.  The table kernel takes the record BY VALUE
// (416 bytes of kernel arguments): pointers that arrive as kernel arguments are known to be global and unclobbered, so the body's
// wave-uniform reads through P / TB stay scalar loads - handed over as a pointer to the record they became 61 flat loads and twice
// the registers (108 instead of 55 for float64 D = 2), and the batch was slower than three chains.
struct DColSweep;
constexpr int kCsBatchMax = 8;
struct DCsBatch {
    const DParams *P[kCsBatchMax];
    const DTabled *TB[kCsBatchMax];
    const DColSweep *CS[kCsBatchMax];      // (the column-sweep kernel's plan: kernels_colsweep.h)
    void *J[kCsBatchMax][2];
    void *idx[kCsBatchMax];
    uint32_t grid[kCsBatchMax];
};
template <typename T, typename TJ, int D>
__global__ void __launch_bounds__(256)
k_backup_tabled32_batch(const DCsBatch B, uint32_t mask, int parity) {
    const unsigned ch = blockIdx.y;
    if (!((mask >> ch) & 1u) || blockIdx.x >= B.grid[ch]) return;
    tabled32_body<T, TJ, D>(B.P[ch], B.TB[ch], (const TJ *)B.J[ch][parity], (TJ *)B.J[ch][parity ^ 1], B.idx[ch], blockIdx.x, B.grid[ch]);
}

}  // namespace hjb
