// kernels_nested.h - variant 1: control-nested stage kernel with cell tracking.
//
// Applies when only the LAST state axis depends on the innermost control dim
// (every spacecraft solver of the reference: Solver_position.m:152-186 v+ = v +
// h*u/M; Solver_attitude.m:423-425 w3+ ... + U3/J3; the C2/C3 synthetic grids).
// Same canonical arithmetic as the generic kernel (bit-identical results), but
//   * everything that does not depend on the innermost control dim - the cells
//     and weights of axes 0..D-2, the partial cost sum - is computed once per
//     OUTER control step instead of once per control;
//   * every axis TRACKS its current cell: the cell's bounds [lo,hi), knot and
//     reciprocal spacing live in registers, so a query that stays inside its cell
//     costs two compares; the exact search runs only on a boundary crossing;
//   * the 2^D-corner gather + the lerps of axes 0..D-2 collapse into a cached
//     pair (E0, dE) per last-axis cell: the inner loop is
//         q = qo + b[j]; t = (q-kc)*rc; val = fma(t,dE,E0); tot = (go+r[j]) + val
//     plus the strict-< min update;
//   * the last axis' knots / reciprocal spacings and the control-only inner
//     tables (b*u, r*u^2), interleaved per control, are staged in LDS once per
//     workgroup (one ds_read_b128 per control brings every inner table value).
#pragma once
#include "hjbdp_dev.h"
#include "kernels_generic.h"

namespace hjb {

constexpr int kMaxInAx = 2;    // inner terms of the last axis
constexpr int kMaxInCost = 2;  // inner terms of the cost
constexpr int kMaxInner = kMaxInAx + kMaxInCost;  // = LDS slots per control (one 16-B row for f32)

struct DInnerTerm {
    const void *data;     // global table
    int32_t stride_in;    // element stride along the innermost control dim
    int32_t lds_slot;     // >= 0: control-only table staged in LDS slot; -1: general
};

// q = r / d for a divisor the host prepared (Granlund - Montgomery, exact for every 32-bit r): see DNested::div_m
__device__ __forceinline__ uint32_t udiv_gm(uint32_t r, uint32_t m, int s) {
    if (s < 0) return r;
    const uint32_t t = __umulhi(r, m);
    return (t + ((r - t) >> 1)) >> s;
}
struct DNested {
    int32_t m_in;         // size of the innermost control dim
    int32_t nUo;          // product of the outer control dims
    int32_t ax_kin;       // first term of the last axis that is evaluated per control
    int32_t cost_kin;     // first cost term that is evaluated per control
    int32_t n_ax_in, n_cost_in;
    int32_t n_slots, pad;
    // Outer control loops: o0 over m_o0 (control dim 0 when C == 3, else 1 trip),
    // o1 over m_o1 (control dim C-2 when C >= 2, else 1 trip).  A term is evaluated
    // at the outermost level that keeps the left-to-right order: terms
    // [n_prefix, l0) once per o0 step, [l0, end) once per (o0,o1) step.
    int32_t m_o0, m_o1;
    int32_t ax_l0[HJB_MAX_D];   // per axis (last axis: end bounded by ax_kin)
    int32_t cost_l0;
    int32_t chunk_order;  // variant 4, window modes: 0 = chunks visited in the transposed order (kernels_packed2.h), 1 = in state order
    // division of a 32-bit state index by the grid sizes n[a] without per-kernel reciprocals (variant 4: the compiler's own
    // expansion keeps one magic number per divisor in a VECTOR register for the whole kernel - and spilled them at five waves):
    // q = (t + ((r - t) >> 1)) >> div_s[a] with t = mulhi(r, div_m[a])  (n[a] == 1: div_m = 0, div_s = 0 gives q = r >> 0 ... see div_by)
    uint32_t div_m[HJB_MAX_D];
    int32_t div_s[HJB_MAX_D];      // -1: divisor 1
    uint32_t div_m_o1, div_m_inner;      // the same for m_o1 (outer control step -> (o0, o1)) and for the states per plane of the last axis
    int32_t div_s_o1, div_s_inner;
    DInnerTerm in[kMaxInner];  // [0,n_ax_in): last-axis terms, [kMaxInAx, kMaxInAx+n_cost_in): cost
    // Variant 2 (packed) only: the canonical shape has at most ONE non-prefix,
    // non-inner term per axis and one cost term per outer level.  ot[a] (a < D):
    // axis a's outer term; ot[HJB_MAX_D], ot[HJB_MAX_D+1]: the cost's level-0 / level-1 term.
    struct DOuterTerm {
        const void *data;
        int32_t sstride[HJB_MAX_D];  // element strides along the state dims
        int32_t c0, c1;              // element strides along the o0 / o1 loop counters
        int32_t present;             // 0: no such term
        int32_t level;               // 0: evaluate once per o0 step, 1: once per (o0,o1) step
        int32_t first;               // 1: it is the first term of its sum (no prefix before it)
        int32_t lds_off;             // >= 0: control-only table staged in LDS at this float offset; -1: global
        int32_t lds_len;             // elements staged
        int32_t pad;
    } ot[HJB_MAX_D + 2];
    // Variant 2: stage-invariant (cell, weight) of every outer axis a < D-1, precomputed at
    // hjb_create over the axis' own broadcast domain (union of its terms' masks) with the
    // canonical arithmetic: entry = {int32 cell, float t}.
    struct DAxisTable {
        const void *tab;             // int2-sized entries
        int32_t sstride[HJB_MAX_D];  // entry strides along the state dims of the domain (0 = not in domain)
        int32_t c0, c1;              // entry strides along the o0 / o1 loop counters
        int32_t level;               // -1: state-only domain, 0: changes per o0 step, 1: per (o0,o1) step
        int32_t pad;
    } at[HJB_MAX_D];
};

// per-axis tracked cell
template <typename T>
struct CellTrack {
    T lo, hi;   // q in [lo, hi) <=> same cell (lo = -inf for cell 0, hi = +inf for cell n-2)
    T kc, rc;   // knot and reciprocal spacing of the cell
    int cell;
};

template <typename T>
__device__ __forceinline__ void track_reset(CellTrack<T> &c) {
    c.lo = (T)INFINITY;
    c.hi = -(T)INFINITY;
    c.kc = (T)0;
    c.rc = (T)0;
    c.cell = 0;
}

// exact: returns true when the cell changed (or on first use)
template <typename T>
__device__ __forceinline__ bool track_update(CellTrack<T> &c, const T *k, const T *r, int n, T q, int uniform, T x0,
                                             T inv_h) {
    if (q >= c.lo && q < c.hi) return false;
    const int i = find_cell<T>(k, n, q, uniform, x0, inv_h);
    c.cell = i;
    c.kc = k[i];
    c.rc = r[i];
    c.lo = (i == 0) ? -(T)INFINITY : c.kc;
    c.hi = (i == n - 2) ? (T)INFINITY : k[i + 1];
    return true;
}

template <typename T, int D>
__device__ __forceinline__ int term_offset(const DTerm &t, const int (&si)[D], const int (&cj)[HJB_MAX_C]) {
    int off = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) off += t.stride[a] * si[a];
#pragma unroll
    for (int c = 0; c < HJB_MAX_C; ++c) off += t.stride[D + c] * cj[c];
    return off;
}

// FAST = the canonical spacecraft shape: exactly one control-only inner term for
// the last axis and one for the cost, each preceded by at least one other term
// (x+ = [state part] + b*u_in, g = [state/outer part] + r*u_in^2).  All the
// runtime structure flags fold away and the inner loop is ~11 VALU per control.
template <typename T, int D, bool FAST>
__global__ void __launch_bounds__(256)
k_backup_nested(const DParams *__restrict__ P, const DNested *__restrict__ N, const T *__restrict__ Jn,
                T *__restrict__ Jout, void *__restrict__ idx_out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const DAxis &axl = P->axis[D - 1];
    const int nl = axl.n;
    const int m_in = N->m_in;
    // LDS: [m_in][kMaxInner] interleaved inner tables (16-B aligned rows), then knots, rdx of the last axis
    T *s_tab = reinterpret_cast<T *>(smem_raw);
    T *s_k = s_tab + (size_t)(m_in + 1) * kMaxInner;  // one padding row (prefetch)
    T *s_r = s_k + nl;
    for (int i = threadIdx.x; i < nl; i += blockDim.x) {
        s_k[i] = static_cast<const T *>(axl.knots)[i];
        s_r[i] = static_cast<const T *>(axl.rdx)[i];
    }
#pragma unroll
    for (int s = 0; s < kMaxInner; ++s) {
        const DInnerTerm &it = N->in[s];
        for (int i = threadIdx.x; i <= m_in; i += blockDim.x)
            s_tab[i * kMaxInner + s] =
                (it.lds_slot >= 0 && i < m_in) ? static_cast<const T *>(it.data)[(int64_t)i * it.stride_in] : (T)0;
    }
    __syncthreads();

    const int C = P->C;
    const int64_t n_owned = P->n_owned;
    const int nUo = N->nUo;
    const int n_ax_in = N->n_ax_in, n_cost_in = N->n_cost_in;
    const int ax_kin = N->ax_kin, cost_kin = N->cost_kin;
    const int l_uniform = axl.uniform;
    const T l_x0 = (T)axl.x0, l_invh = (T)axl.inv_h;
    const int plane0 = P->plane0, nplanes = P->nplanes;
    const int64_t js_last = P->jstride[D - 1];
    // which inner slots are general (need a per-lane global load)?
    bool gen[kMaxInner];
    const T *gdata[kMaxInner];
    int gstride[kMaxInner];
#pragma unroll
    for (int s = 0; s < kMaxInner; ++s) {
        const bool used = (s < kMaxInAx) ? (s < n_ax_in) : (s - kMaxInAx < n_cost_in);
        gen[s] = used && N->in[s].lds_slot < 0;
        gdata[s] = static_cast<const T *>(N->in[s].data);
        gstride[s] = N->in[s].stride_in;
    }
    const bool any_gen = gen[0] || gen[1] || gen[2] || gen[3];

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
        CellTrack<T> trk[D];
#pragma unroll
        for (int a = 0; a < D; ++a) track_reset(trk[a]);
        int lc = 0;  // local plane of the tracked last-axis cell

        T best = (T)0;
        int best_uo = 0, best_j = 0;
        const int m_o0 = N->m_o0, m_o1 = N->m_o1;
        int uo = 0;
        for (int o0 = 0; o0 < m_o0; ++o0) {
            if (C == 3) cj[0] = o0;
            // ---- level 0: once per step of the outermost control dim ----------
            T tw[D > 1 ? D - 1 : 1];
            T q0[D > 1 ? D - 1 : 1];
#pragma unroll
            for (int a = 0; a < D - 1; ++a) {
                const DAxis &ax = P->axis[a];
                T q = qpre[a];
                for (int k = ax.n_prefix; k < N->ax_l0[a]; ++k) {
                    T x = term_value<T, D>(ax.t[k], si, cj);
                    q = (k == 0) ? x : (T)(q + x);
                }
                q0[a] = q;
                if (N->ax_l0[a] == ax.n_terms) {   // fully resolved at this level
                    track_update<T>(trk[a], static_cast<const T *>(ax.knots), static_cast<const T *>(ax.rdx), ax.n, q,
                                    ax.uniform, (T)ax.x0, (T)ax.inv_h);
                    tw[a] = (T)((T)(q - trk[a].kc) * trk[a].rc);
                }
            }
            T qo0 = qpre[D - 1];
            for (int k = axl.n_prefix; k < N->ax_l0[D - 1]; ++k) {
                T x = term_value<T, D>(axl.t[k], si, cj);
                qo0 = (k == 0) ? x : (T)(qo0 + x);
            }
            T go0 = gpre;
            for (int k = P->n_cost_prefix; k < N->cost_l0; ++k) {
                T x = term_value<T, D>(P->cost[k], si, cj);
                go0 = (k == 0) ? x : (T)(go0 + x);
            }
          for (int o1 = 0; o1 < m_o1; ++o1, ++uo) {
            if (C == 3) cj[1] = o1; else if (C == 2) cj[0] = o1;
            // ---- level 1: once per outer control step --------------------------
            int64_t base = 0;
#pragma unroll
            for (int a = 0; a < D - 1; ++a) {
                const DAxis &ax = P->axis[a];
                if (N->ax_l0[a] != ax.n_terms) {
                    T q = q0[a];
                    for (int k = N->ax_l0[a]; k < ax.n_terms; ++k) {
                        T x = term_value<T, D>(ax.t[k], si, cj);
                        q = (k == 0) ? x : (T)(q + x);
                    }
                    track_update<T>(trk[a], static_cast<const T *>(ax.knots), static_cast<const T *>(ax.rdx), ax.n, q,
                                    ax.uniform, (T)ax.x0, (T)ax.inv_h);
                    tw[a] = (T)((T)(q - trk[a].kc) * trk[a].rc);
                }
                base += P->jstride[a] * trk[a].cell;
            }
            T qo = qo0;
            for (int k = N->ax_l0[D - 1]; k < ax_kin; ++k) {
                T x = term_value<T, D>(axl.t[k], si, cj);
                qo = (k == 0) ? x : (T)(qo + x);
            }
            T go = go0;
            for (int k = N->cost_l0; k < cost_kin; ++k) {
                T x = term_value<T, D>(P->cost[k], si, cj);
                go = (k == 0) ? x : (T)(go + x);
            }
            int ib[kMaxInner] = {0, 0, 0, 0};
            if (any_gen) {
#pragma unroll
                for (int s = 0; s < kMaxInner; ++s)
                    if (gen[s])
                        ib[s] = term_offset<T, D>(s < kMaxInAx ? axl.t[ax_kin + s] : P->cost[cost_kin + s - kMaxInAx],
                                                  si, cj);
            }
            bool need_gather = true;
            T E0 = (T)0, dE = (T)0;
            // ---- per control of the innermost dim ----------------------------
            if constexpr (FAST) {
                // slot 0 = last-axis table b[j], slot kMaxInAx = cost table r[j]
                T xb = s_tab[0], xr = s_tab[kMaxInAx];
                T ibest = (T)INFINITY;
                int ij = 0;
                for (int j = 0; j < m_in; ++j) {
                    const T q = (T)(qo + xb);
                    const T gr = (T)(go + xr);
                    // prefetch the next control's row (row m_in is padding)
                    xb = s_tab[(j + 1) * kMaxInner];
                    xr = s_tab[(j + 1) * kMaxInner + kMaxInAx];
                    if (track_update<T>(trk[D - 1], s_k, s_r, nl, q, l_uniform, l_x0, l_invh)) {
                        lc = trk[D - 1].cell - plane0;
                        if (lc < 0 || lc + 1 >= nplanes) {
                            *P->status = 1;
                            lc = lc < 0 ? 0 : nplanes - 2;
                        }
                        need_gather = true;
                    }
                    if (need_gather) {
                        need_gather = false;
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
                    const T t = (T)((T)(q - trk[D - 1].kc) * trk[D - 1].rc);
                    const T tot = (T)(gr + fma_t<T>(t, dE, E0));
                    if (tot < ibest) {
                        ibest = tot;
                        ij = j;
                    }
                }
                // strict '<' across outer steps keeps the first minimiser in visiting order
                if (uo == 0 || ibest < best) {
                    best = ibest;
                    best_uo = uo;
                    best_j = ij;
                }
            } else {
                for (int j = 0; j < m_in; ++j) {
                    T x[kMaxInner];
    #pragma unroll
                    for (int s = 0; s < kMaxInner; ++s) x[s] = s_tab[j * kMaxInner + s];
                    if (any_gen) {
    #pragma unroll
                        for (int s = 0; s < kMaxInner; ++s)
                            if (gen[s]) x[s] = gdata[s][ib[s] + j * gstride[s]];
                    }
                    T q = qo;
                    if (n_ax_in > 0) q = (ax_kin == 0) ? x[0] : (T)(q + x[0]);
                    if (n_ax_in > 1) q = (T)(q + x[1]);
                    if (track_update<T>(trk[D - 1], s_k, s_r, nl, q, l_uniform, l_x0, l_invh)) {
                        lc = trk[D - 1].cell - plane0;
                        if (lc < 0 || lc + 1 >= nplanes) {
                            *P->status = 1;
                            lc = lc < 0 ? 0 : nplanes - 2;
                        }
                        need_gather = true;
                    }
                    if (need_gather) {
                        need_gather = false;
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
                    const T t = (T)((T)(q - trk[D - 1].kc) * trk[D - 1].rc);
                    const T val = fma_t<T>(t, dE, E0);
                    T g = go;
                    if (n_cost_in > 0) g = (cost_kin == 0) ? x[kMaxInAx] : (T)(g + x[kMaxInAx]);
                    if (n_cost_in > 1) g = (T)(g + x[kMaxInAx + 1]);
                    const T tot = (T)(g + val);
                    if ((uo == 0 && j == 0) || tot < best) {
                        best = tot;
                        best_uo = uo;
                        best_j = j;
                    }
                }
            }
          }  // o1
        }  // o0
        // visiting order (dim 0 slowest) -> column-major label (dim 0 fastest)
        int64_t label;
        if (C == 1) {
            label = best_j;
        } else if (C == 2) {
            label = best_uo + (int64_t)P->m[0] * best_j;
        } else {
            const int j1 = best_uo % P->m[1], j0 = best_uo / P->m[1];
            label = j0 + (int64_t)P->m[0] * (j1 + (int64_t)P->m[1] * best_j);
        }
        const int64_t in_plane = ls % P->inner, pl = ls / P->inner;
        Jout[in_plane + P->inner * (pl + P->halo_lo)] = best;
        if (idx_out) st_idx(idx_out, ls, (int32_t)(label + P->index_base), P->idx_bytes);
    }
}

}  // namespace hjb
