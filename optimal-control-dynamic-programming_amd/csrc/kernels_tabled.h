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
    for (int64_t ls = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; ls < n_owned;
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

}  // namespace hjb
