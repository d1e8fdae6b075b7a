// kernels_nested.h - variant 1: control-nested stage kernel.
//
// Applies when only the LAST state axis depends on the innermost control dim
// (every spacecraft solver of the reference: Solver_position.m:152-186 v+ = v +
// h*u/M; Solver_attitude.m:423-425 w3+ ... + U3/J3; the C2/C3 synthetic grids).
// Same canonical arithmetic as the generic kernel (bit-identical results), but
//   * everything that does not depend on the innermost control dim - the cells
//     and weights of axes 0..D-2, the partial cost sum - is computed once per
//     OUTER control step instead of once per control;
//   * the 2^D-corner gather + the lerps of axes 0..D-2 collapse into a cached
//     pair (E0, dE) tagged with the last-axis cell; the inner loop is one fma
//     per control unless the query crosses a cell boundary;
//   * the last axis' knots / reciprocal spacings and the control-only inner
//     tables (b*u, r*u^2) are staged in LDS once per workgroup.
// Per control: 1 LDS table read + cell search on LDS knots + fma + 2 adds +
// compare/select, instead of D searches + 2^D global loads + (2^D-1) lerps.
#pragma once
#include "hjbdp_dev.h"
#include "kernels_generic.h"

namespace hjb {

constexpr int kMaxInAx = 2;    // inner terms of the last axis kept in registers
constexpr int kMaxInCost = 3;  // inner terms of the cost
constexpr int kMaxInner = kMaxInAx + kMaxInCost;

struct DInnerTerm {
    const void *data;     // global table
    int32_t stride_in;    // element stride along the innermost control dim
    int32_t lds_slot;     // >= 0: control-only table staged in LDS slot; -1: general
};

struct DNested {
    int32_t m_in;         // size of the innermost control dim
    int32_t nUo;          // product of the outer control dims
    int32_t ax_kin;       // first term of the last axis that is evaluated per control
    int32_t cost_kin;     // first cost term that is evaluated per control
    int32_t n_ax_in, n_cost_in;
    int32_t n_slots, pad;
    DInnerTerm in[kMaxInner];  // [0,n_ax_in): last-axis terms, [kMaxInAx, kMaxInAx+n_cost_in): cost
};

template <typename T>
__device__ __forceinline__ int find_cell_lds(const T *k, int n, T q, int uniform, T x0, T inv_h) {
    return find_cell<T>(k, n, q, uniform, x0, inv_h);
}

template <typename T, int D>
__global__ void __launch_bounds__(256)
k_backup_nested(const DParams *__restrict__ P, const DNested *__restrict__ N, const T *__restrict__ Jn,
                T *__restrict__ Jout, int32_t *__restrict__ idx_out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const DAxis &axl = P->axis[D - 1];
    const int nl = axl.n;
    const int m_in = N->m_in;
    T *s_k = reinterpret_cast<T *>(smem_raw);
    T *s_r = s_k + nl;
    T *s_tab = s_r + nl;
    for (int i = threadIdx.x; i < nl; i += blockDim.x) {
        s_k[i] = static_cast<const T *>(axl.knots)[i];
        s_r[i] = static_cast<const T *>(axl.rdx)[i];
    }
#pragma unroll
    for (int s = 0; s < kMaxInner; ++s) {
        const DInnerTerm &it = N->in[s];
        if (it.lds_slot >= 0)
            for (int i = threadIdx.x; i < m_in; i += blockDim.x)
                s_tab[it.lds_slot * m_in + i] = static_cast<const T *>(it.data)[(int64_t)i * it.stride_in];
    }
    __syncthreads();

    const int C = P->C;
    const int64_t n_owned = P->n_owned;
    const int nUo = N->nUo;
    const int n_ax_in = N->n_ax_in, n_cost_in = N->n_cost_in;
    const int l_uniform = axl.uniform;
    const T l_x0 = (T)axl.x0, l_invh = (T)axl.inv_h;
    const int plane0 = P->plane0, nplanes = P->nplanes;
    const int64_t js_last = P->jstride[D - 1];

    for (int64_t ls = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; ls < n_owned;
         ls += (int64_t)gridDim.x * blockDim.x) {
        int si[D];
        {
            int64_t r = ls;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                int na = P->n[a];
                si[a] = (int)(r % na);
                r /= na;
            }
            si[D - 1] += P->slab_begin;
        }
        int cj[HJB_MAX_C] = {0, 0, 0};
        T qpre[D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const DAxis &ax = P->axis[a];
            T q = (T)0;
            for (int k = 0; k < ax.n_prefix; ++k) {
                T x = term_value<T, D>(ax.t[k], si, cj);
                q = (k == 0) ? x : (T)(q + x);
            }
            qpre[a] = q;
        }
        T gpre = (T)0;
        for (int k = 0; k < P->n_cost_prefix; ++k) {
            T x = term_value<T, D>(P->cost[k], si, cj);
            gpre = (k == 0) ? x : (T)(gpre + x);
        }

        T best = (T)0;
        int64_t best_u = 0;
        for (int uo = 0; uo < nUo; ++uo) {
            // ---- once per outer control step -------------------------------
            T tw[D > 1 ? D - 1 : 1];
            int64_t base = 0;
#pragma unroll
            for (int a = 0; a < D - 1; ++a) {
                const DAxis &ax = P->axis[a];
                T q = qpre[a];
                for (int k = ax.n_prefix; k < ax.n_terms; ++k) {
                    T x = term_value<T, D>(ax.t[k], si, cj);
                    q = (k == 0) ? x : (T)(q + x);
                }
                const T *kk = static_cast<const T *>(ax.knots);
                int cell = find_cell<T>(kk, ax.n, q, ax.uniform, (T)ax.x0, (T)ax.inv_h);
                tw[a] = (T)((T)(q - kk[cell]) * static_cast<const T *>(ax.rdx)[cell]);
                base += P->jstride[a] * cell;
            }
            T qo = qpre[D - 1];
            for (int k = axl.n_prefix; k < N->ax_kin; ++k) {
                T x = term_value<T, D>(axl.t[k], si, cj);
                qo = (k == 0) ? x : (T)(qo + x);
            }
            T go = gpre;
            for (int k = P->n_cost_prefix; k < N->cost_kin; ++k) {
                T x = term_value<T, D>(P->cost[k], si, cj);
                go = (k == 0) ? x : (T)(go + x);
            }
            // base offsets of the general (non control-only) inner terms
            int64_t ib[kMaxInner];
#pragma unroll
            for (int s = 0; s < kMaxInAx; ++s) {
                ib[s] = 0;
                if (s < n_ax_in && N->in[s].lds_slot < 0) {
                    const DTerm &t = axl.t[N->ax_kin + s];
                    int64_t off = 0;
#pragma unroll
                    for (int a = 0; a < D; ++a) off += (int64_t)t.stride[a] * si[a];
#pragma unroll
                    for (int c = 0; c < HJB_MAX_C; ++c) off += (int64_t)t.stride[D + c] * cj[c];
                    ib[s] = off;
                }
            }
#pragma unroll
            for (int s = 0; s < kMaxInCost; ++s) {
                ib[kMaxInAx + s] = 0;
                if (s < n_cost_in && N->in[kMaxInAx + s].lds_slot < 0) {
                    const DTerm &t = P->cost[N->cost_kin + s];
                    int64_t off = 0;
#pragma unroll
                    for (int a = 0; a < D; ++a) off += (int64_t)t.stride[a] * si[a];
#pragma unroll
                    for (int c = 0; c < HJB_MAX_C; ++c) off += (int64_t)t.stride[D + c] * cj[c];
                    ib[kMaxInAx + s] = off;
                }
            }
            int tag = -0x7fffffff;
            T E0 = (T)0, dE = (T)0;
            // ---- per control of the innermost dim ----------------------------
            for (int j = 0; j < m_in; ++j) {
                T q = qo;
#pragma unroll
                for (int s = 0; s < kMaxInAx; ++s) {
                    if (s < n_ax_in) {
                        const DInnerTerm &it = N->in[s];
                        T x = it.lds_slot >= 0
                                  ? s_tab[it.lds_slot * m_in + j]
                                  : static_cast<const T *>(it.data)[ib[s] + (int64_t)j * it.stride_in];
                        q = (N->ax_kin + s == 0) ? x : (T)(q + x);
                    }
                }
                int cell = find_cell_lds<T>(s_k, nl, q, l_uniform, l_x0, l_invh);
                const T t = (T)((T)(q - s_k[cell]) * s_r[cell]);
                int lc = cell - plane0;
                if (lc < 0 || lc + 1 >= nplanes) {
                    *P->status = 1;
                    lc = lc < 0 ? 0 : nplanes - 2;
                }
                if (lc != tag) {
                    tag = lc;
                    T v[1 << D];
                    const int64_t b2 = base + js_last * lc;
#pragma unroll
                    for (int c = 0; c < (1 << D); ++c) {
                        int64_t off = b2;
#pragma unroll
                        for (int a = 0; a < D; ++a)
                            if (c & (1 << a)) off += P->jstride[a];
                        v[c] = Jn[off];
                    }
#pragma unroll
                    for (int a = 0; a < D - 1; ++a) {
#pragma unroll
                        for (int jj = 0; jj < (1 << (D - 1 - a)); ++jj)
                            v[jj] = fma_t<T>(tw[a], (T)(v[2 * jj + 1] - v[2 * jj]), v[2 * jj]);
                    }
                    E0 = v[0];
                    dE = (T)(v[1] - v[0]);
                }
                const T val = fma_t<T>(t, dE, E0);
                T g = go;
#pragma unroll
                for (int s = 0; s < kMaxInCost; ++s) {
                    if (s < n_cost_in) {
                        const DInnerTerm &it = N->in[kMaxInAx + s];
                        T x = it.lds_slot >= 0
                                  ? s_tab[it.lds_slot * m_in + j]
                                  : static_cast<const T *>(it.data)[ib[kMaxInAx + s] + (int64_t)j * it.stride_in];
                        g = (N->cost_kin + s == 0) ? x : (T)(g + x);
                    }
                }
                const T tot = (T)(g + val);
                const int64_t u = (int64_t)uo * m_in + j;
                if (u == 0 || tot < best) {
                    best = tot;
                    best_u = u;
                }
            }
            // next outer control: dims 0..C-2, dim C-2 fastest
            if (C == 3) {
                if (++cj[1] == P->m[1]) { cj[1] = 0; ++cj[0]; }
            } else if (C == 2) {
                ++cj[0];
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
        Jout[in_plane + P->inner * (pl + P->halo_lo)] = best;
        if (idx_out) idx_out[ls] = (int32_t)(label + P->index_base);
    }
}

}  // namespace hjb
