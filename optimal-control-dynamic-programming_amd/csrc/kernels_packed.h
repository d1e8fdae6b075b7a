// kernels_packed.h - variant 2: control-nested kernel, two states per lane,
// packed fp32 math, precomputed stage-invariant interpolation tables.
//
// gfx950 executes a wave64 VALU instruction in 4 cycles and reaches its fp32 peak
// only with the packed forms (v_pk_add/mul/fma_f32); rocprof showed variant 1
// ~100 % VALU-issue bound (23 VALU + 18 SALU per control).  This kernel handles
// the canonical spacecraft shape
//      x+_last = S(state) + b[u_in]          (no outer-control term on the last axis)
//      x+_a    = anything not depending on u_in,            a < D-1
//      g       = G(state) [+ c0(u_0...)] [+ c1(u_0,u_1...)] + r[u_in]
// (Solver_position, the C2 grid, ...; decided on the host) and exploits that all
// interpolation cells and weights are STAGE-INVARIANT:
//   * for every outer axis a < D-1 the pair (cell, t) is precomputed once, at
//     hjb_create, over the axis' own broadcast domain (k_prep_axis_table, same
//     canonical arithmetic); a loop level costs one 8-byte lookup per axis;
//   * for the last axis q_j = qpre + b[j] does not depend on the outer controls:
//     per state the weights t_j go to LDS once ([j][lane] float2 for the lane's
//     two states) and the controls at which the query enters a new cell become a
//     per-lane bit mask, OR-ed across the wave into a scalar mask - the hot loop
//     tests cell changes with scalar instructions only;
//   * every lane carries TWO states (s, s + 256) in float2 registers: per control
//     the hot loop is  g = go + r[j];  tot = g + fma(t_j, dE, E0)  in packed
//     instructions plus the strict-< argmin update, LDS rows prefetched;
//   * the 2^D corner loads of outer step o1+1 are issued before the inner loop
//     of step o1 (software pipeline); 32-bit offsets, paired 8-byte corner loads.
// Arithmetic is exactly the canonical order: results are bit-identical to
// variants 0/1 and to the CPU oracle twin.
#pragma once
#include "hjbdp_dev.h"
#include "kernels_nested.h"

namespace hjb {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

template <int D>
__device__ __forceinline__ float term_value32(const DTerm &t, const int (&si)[D]) {
    int off = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) off += t.stride[a] * si[a];
    return as_global<float>(t.data)[off];   // state-only (prefix) terms
}

// same, J stored as IEEE half: two adjacent halves per 4-byte load
typedef _Float16 h2u __attribute__((ext_vector_type(2), aligned(2)));
template <int D>
__device__ __forceinline__ void load_corners(const _Float16 *__restrict__ Jn, int b2, const int (&js)[D],
                                             float (&v)[1 << D]) {
    if constexpr (D >= 2) {
#pragma unroll
        for (int c = 0; c < (1 << D); c += 2) {
            int off = b2;
#pragma unroll
            for (int a = 1; a < D; ++a)
                if (c & (1 << a)) off += js[a];
            const h2u p = *reinterpret_cast<const h2u *>(Jn + off);
            v[c] = (float)p.x;
            v[c + 1] = (float)p.y;
        }
    } else {
        v[0] = (float)Jn[b2];
        v[1] = (float)Jn[b2 + js[0]];
    }
}

template <int D>
__device__ __forceinline__ void load_corners(const float *__restrict__ Jn, int b2, const int (&js)[D],
                                             float (&v)[1 << D]) {
    if constexpr (D >= 2) {
#pragma unroll
        for (int c = 0; c < (1 << D); c += 2) {
            int off = b2;
#pragma unroll
            for (int a = 1; a < D; ++a)
                if (c & (1 << a)) off += js[a];
            const f2u p = *reinterpret_cast<const f2u *>(Jn + off);  // js[0] == 1: corners c, c+1 are adjacent
            v[c] = p.x;
            v[c + 1] = p.y;
        }
    } else {
        v[0] = Jn[b2];
        v[1] = Jn[b2 + js[0]];
    }
}

// the two axis-0 neighbours at element offset off (axis 0 has stride 1)
__device__ __forceinline__ void load_pair(const float *__restrict__ Jn, int off, float (&v)[2]) {
    const f2u p = *reinterpret_cast<const f2u *>(Jn + off);
    v[0] = p.x;
    v[1] = p.y;
}
__device__ __forceinline__ void load_pair(const _Float16 *__restrict__ Jn, int off, float (&v)[2]) {
    const h2u p = *reinterpret_cast<const h2u *>(Jn + off);
    v[0] = (float)p.x;
    v[1] = (float)p.y;
}

// contract axes 0..D-2 with weights tw[] in canonical order -> E0 and dE = E1 - E0
template <int D>
__device__ __forceinline__ void contract(float (&v)[1 << D], const float (&tw)[D > 1 ? D - 1 : 1], float &E0, float &dE) {
#pragma unroll
    for (int a = 0; a < D - 1; ++a) {
#pragma unroll
        for (int jj = 0; jj < (1 << (D - 1 - a)); ++jj)
            v[jj] = __builtin_fmaf(tw[a], v[2 * jj + 1] - v[2 * jj], v[2 * jj]);
    }
    E0 = v[0];
    dE = v[1] - v[0];
}

constexpr int kPackedMaxIn = 32;   // crossing masks are 32-bit
// Unrolling the hot loop x3 saves scalar/branch issue but costs ~20 VGPRs (2 instead of 3 waves/SIMD).
constexpr bool kPackedUnroll3 = false;

template <int D>
__global__ void __launch_bounds__(256)
k_backup_packed(const DParams *__restrict__ P, const DNested *__restrict__ N, const float *__restrict__ Jn,
                float *__restrict__ Jout, void *__restrict__ idx_out) {
    constexpr int DM = D > 1 ? D - 1 : 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const DAxis &axl = P->axis[D - 1];
    const int nl = axl.n;
    const int m_in = N->m_in;
    // LDS: t_j of both states per lane [m_in+1][256] float2 | {b[j], r[j]} [m_in+1] (last rows = prefetch
    // padding) | knots, rdx of the last axis | control-only cost tables
    f2 *s_t = reinterpret_cast<f2 *>(smem_raw);
    f2 *s_br = s_t + (size_t)(m_in + 1) * 256;
    float *s_k = reinterpret_cast<float *>(s_br + (m_in + 1));
    float *s_r = s_k + nl;
    float *s_ot = s_r + nl;
    for (int i = threadIdx.x; i < nl; i += blockDim.x) {
        s_k[i] = static_cast<const float *>(axl.knots)[i];
        s_r[i] = static_cast<const float *>(axl.rdx)[i];
    }
    {
        const DInnerTerm &tb = N->in[0];
        const DInnerTerm &tr = N->in[kMaxInAx];
        for (int i = threadIdx.x; i <= m_in; i += blockDim.x) {
            f2 x = {0.f, 0.f};
            if (i < m_in) {
                x.x = static_cast<const float *>(tb.data)[i * tb.stride_in];
                x.y = static_cast<const float *>(tr.data)[i * tr.stride_in];
            }
            s_br[i] = x;
        }
        s_t[(size_t)m_in * 256 + threadIdx.x] = (f2){0.f, 0.f};
    }
    constexpr int CL0 = HJB_MAX_D, CL1 = HJB_MAX_D + 1;
#pragma unroll
    for (int i = CL0; i <= CL1; ++i) {
        const auto &t = N->ot[i];
        if (t.present && t.lds_off >= 0)
            for (int e = threadIdx.x; e < t.lds_len; e += blockDim.x)
                s_ot[t.lds_off + e] = static_cast<const float *>(t.data)[e];
    }
    __syncthreads();

    const int C = P->C;
    const int n_owned = (int)P->n_owned;
    const int l_uniform = axl.uniform;
    const float l_x0 = (float)axl.x0, l_invh = (float)axl.inv_h;
    const int plane0 = P->plane0, nplanes = P->nplanes;
    const int m_o0 = N->m_o0, m_o1 = N->m_o1;
    int js[D];
#pragma unroll
    for (int a = 0; a < D; ++a) js[a] = (int)P->jstride[a];
    const int inner_sz = (int)P->inner;
    f2 *my_t = s_t + threadIdx.x;
    // per-axis table descriptors in registers
    const int2 *atab[DM];
    int a_c0[DM], a_c1[DM], a_lvl[DM];
#pragma unroll
    for (int a = 0; a < D - 1; ++a) {
        atab[a] = static_cast<const int2 *>(N->at[a].tab);
        a_c0[a] = N->at[a].c0;
        a_c1[a] = N->at[a].c1;
        a_lvl[a] = N->at[a].level;
    }
    const bool cl0_present = N->ot[CL0].present, cl1_present = N->ot[CL1].present;
    const bool cl0_first = N->ot[CL0].first, cl1_first = N->ot[CL1].first;

    // each workgroup pass covers 512 consecutive states: lane -> s and s + 256
    for (int blk = blockIdx.x * 512; blk < n_owned; blk += gridDim.x * 512) {
        int ls[2];
        bool valid[2];
        ls[0] = blk + threadIdx.x;
        ls[1] = blk + 256 + threadIdx.x;
        valid[0] = ls[0] < n_owned;
        valid[1] = ls[1] < n_owned;
        if (!valid[0]) ls[0] = n_owned - 1;   // harmless duplicate work, store skipped
        if (!valid[1]) ls[1] = ls[0];

        float ql[2], gpre[2];                 // last-axis state part, cost state part
        int aoff[2][DM];                      // state part of each axis-table offset
        int coff[2][2];                       // state part of the cost level-0 / level-1 term offsets
        int cell[2][DM];
        float tw[2][DM];
        int lc0[2];                           // local plane of the last-axis cell at control 0
        int lc1[2];                           // ... and after this state's FIRST cell change (= lc0 if none)
        unsigned int cm[2] = {0u, 0u};        // bit j: this state's last-axis cell changes at control j
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            int si[D];                        // GLOBAL grid indices (tables of terms are global)
            int r = ls[s];
#pragma unroll
            for (int a = 0; a < D; ++a) {
                int na = P->n[a];
                si[a] = r % na;
                r /= na;
            }
            const int last_local = si[D - 1];
            si[D - 1] += P->slab_begin;
            {
                float q = 0.f;
                for (int k = 0; k < axl.n_prefix; ++k) {
                    float x = term_value32<D>(axl.t[k], si);
                    q = (k == 0) ? x : q + x;
                }
                ql[s] = q;
                float g = 0.f;
                for (int k = 0; k < P->n_cost_prefix; ++k) {
                    float x = term_value32<D>(P->cost[k], si);
                    g = (k == 0) ? x : g + x;
                }
                gpre[s] = g;
            }
#pragma unroll
            for (int a = 0; a < D - 1; ++a) {
                int off = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) off += N->at[a].sstride[d] * (d == D - 1 ? last_local : si[d]);
                aoff[s][a] = off;
                if (a_lvl[a] < 0) {           // state-only domain: resolved once per state
                    const int2 e = atab[a][off];
                    cell[s][a] = e.x;
                    tw[s][a] = __int_as_float(e.y);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const auto &t = N->ot[CL0 + i];
                int off = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) off += t.sstride[d] * si[d];
                coff[s][i] = off;
            }
        }
        // ---- once per state pair: inner weights t_j and cell-crossing bits ------
        {
            CellTrack<float> tl[2];
            track_reset(tl[0]);
            track_reset(tl[1]);
            for (int j = 0; j < m_in; ++j) {
                const float bj = s_br[j].x;
                f2 t;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float q = ql[s] + bj;
                    const bool ch = track_update<float>(tl[s], s_k, s_r, nl, q, l_uniform, l_x0, l_invh);
                    if (j == 0) {
                        lc0[s] = lc1[s] = tl[s].cell - plane0;
                    } else if (ch) {
                        if (cm[s] == 0u) lc1[s] = tl[s].cell - plane0;
                        cm[s] |= 1u << j;
                    }
                    t[s] = (q - tl[s].kc) * tl[s].rc;
                }
                my_t[j * 256] = t;
            }
        }
        // wave-uniform union of the crossing bits (lanes only read their own s_t column: no barrier needed)
        unsigned int U = 0u;
        for (int j = 1; j < m_in; ++j) {
            const bool mine = ((cm[0] | cm[1]) >> j) & 1u;
            if (__ballot(mine)) U |= 1u << j;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (lc0[s] < 0 || lc0[s] + 1 >= nplanes) {
                *P->status = 1;
                lc0[s] = lc0[s] < 0 ? 0 : nplanes - 2;
            }
            if (lc1[s] < 0 || lc1[s] + 1 >= nplanes) {
                *P->status = 1;
                lc1[s] = lc1[s] < 0 ? 0 : nplanes - 2;
            }
        }
        // a state whose query changes cell more than once per inner sweep takes the general (search +
        // gather) path at its 2nd, 3rd ... crossing; wave-uniform flag
        const bool multi = __ballot((cm[0] & (cm[0] - 1u)) != 0u || (cm[1] & (cm[1] - 1u)) != 0u) != 0ull;
        float best[2] = {0.f, 0.f};
        int best_uo[2] = {0, 0}, best_j[2] = {0, 0};

        // a cost term of an outer level at (state s, loop counters o0,o1)
        auto cterm = [&](int slot, int s, int o0, int o1) -> float {
            const auto &t = N->ot[slot];
            if (t.lds_off >= 0) return s_ot[t.lds_off + o0 * t.c0 + o1 * t.c1];   // control-only: wave-uniform
            return static_cast<const float *>(t.data)[coff[s][slot - CL0] + o0 * t.c0 + o1 * t.c1];
        };

        int uo = 0;
        for (int o0 = 0; o0 < m_o0; ++o0) {
            // ---- level 0: once per step of the outermost control dim ----------
            float go0[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int a = 0; a < D - 1; ++a) {
                    if (a_lvl[a] == 0) {
                        const int2 e = atab[a][aoff[s][a] + o0 * a_c0[a]];
                        cell[s][a] = e.x;
                        tw[s][a] = __int_as_float(e.y);
                    }
                }
                go0[s] = gpre[s];
                if (cl0_present) {
                    const float x = cterm(CL0, s, o0, 0);
                    go0[s] = cl0_first ? x : go0[s] + x;
                }
            }
            // level-1 work for step (o0,o1): table lookups of the level-1 axes, J base offset, cost so far
            auto prepare = [&](int o1, int (&base)[2], float (&twc)[2][DM], f2 &go) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    int b = 0;
#pragma unroll
                    for (int a = 0; a < D - 1; ++a) {
                        if (a_lvl[a] == 1) {
                            const int2 e = atab[a][aoff[s][a] + o0 * a_c0[a] + o1 * a_c1[a]];
                            cell[s][a] = e.x;
                            tw[s][a] = __int_as_float(e.y);
                        }
                        b += js[a] * cell[s][a];
                        twc[s][a] = tw[s][a];
                    }
                    base[s] = b;
                    float g = go0[s];
                    if (cl1_present) {
                        const float x = cterm(CL1, s, o0, o1);
                        g = cl1_first ? x : g + x;
                    }
                    go[s] = g;
                }
            };
            int base[2], base_n[2];
            float twc[2][DM], twn[2][DM];
            f2 go, go_n;
            float G[2][1 << D], H[2][1 << D];   // corners around the first / second last-axis cell
            prepare(0, base, twc, go);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                load_corners<D>(Jn, base[s] + js[D - 1] * lc0[s], js, G[s]);
                load_corners<D>(Jn, base[s] + js[D - 1] * lc1[s], js, H[s]);
            }
            for (int o1 = 0; o1 < m_o1; ++o1, ++uo) {
                // ---- level 1: contract the prefetched corners of this step ---------
                f2 E0, dE, E0b, dEb;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float e0, de;
                    contract<D>(G[s], twc[s], e0, de);
                    E0[s] = e0;
                    dE[s] = de;
                    contract<D>(H[s], twc[s], e0, de);
                    E0b[s] = e0;
                    dEb[s] = de;
                }
                // ---- issue the next step's corner loads; they land during the inner loop
                const bool has_next = o1 + 1 < m_o1;
                if (has_next) {
                    prepare(o1 + 1, base_n, twn, go_n);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        load_corners<D>(Jn, base_n[s] + js[D - 1] * lc0[s], js, G[s]);
                        load_corners<D>(Jn, base_n[s] + js[D - 1] * lc1[s], js, H[s]);
                    }
                }
                float ibest[2] = {INFINITY, INFINITY};
                int ij[2] = {0, 0};
                // ---- inner loop: packed, no cell tests and no loads on the vector pipe ------
                f2 t = my_t[0];
                float rj = s_br[0].y;
                auto cross = [&](int j) {                        // executed only where some lane changes cell
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const unsigned int first = cm[s] & (0u - cm[s]);   // lowest set bit = first crossing
                        if ((first >> j) & 1u) {                 // first crossing: corners were prefetched
                            E0[s] = E0b[s];
                            dE[s] = dEb[s];
                        } else if (multi && ((cm[s] >> j) & 1u)) {     // later crossings: general path
                            const float q = ql[s] + s_br[j].x;
                            int lc = find_cell<float>(s_k, nl, q, l_uniform, l_x0, l_invh) - plane0;
                            if (lc < 0 || lc + 1 >= nplanes) {
                                *P->status = 1;
                                lc = lc < 0 ? 0 : nplanes - 2;
                            }
                            float v[1 << D], e0, de;
                            load_corners<D>(Jn, base[s] + js[D - 1] * lc, js, v);
                            contract<D>(v, twc[s], e0, de);
                            E0[s] = e0;
                            dE[s] = de;
                        }
                    }
                };
                auto body = [&](int j) {
                    const f2 g = go + (f2){rj, rj};
                    const f2 tot = g + __builtin_elementwise_fma(t, dE, E0);
                    t = my_t[(j + 1) * 256];                     // next control's rows (row m_in is padding)
                    rj = s_br[j + 1].y;
                    if (tot.x < ibest[0]) { ibest[0] = tot.x; ij[0] = j; }
                    if (tot.y < ibest[1]) { ibest[1] = tot.y; ij[1] = j; }
                };
                int j = 0;
                for (; kPackedUnroll3 && j + 3 <= m_in; j += 3) {
                    if ((U >> j) & 7u) {                         // scalar test for the whole group
                        if ((U >> j) & 1u) cross(j);
                        body(j);
                        if ((U >> (j + 1)) & 1u) cross(j + 1);
                        body(j + 1);
                        if ((U >> (j + 2)) & 1u) cross(j + 2);
                        body(j + 2);
                    } else {
                        body(j);
                        body(j + 1);
                        body(j + 2);
                    }
                }
                for (; j < m_in; ++j) {
                    if ((U >> j) & 1u) cross(j);
                    body(j);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (uo == 0 || ibest[s] < best[s]) {
                        best[s] = ibest[s];
                        best_uo[s] = uo;
                        best_j[s] = ij[s];
                    }
                }
                if (has_next) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        base[s] = base_n[s];
#pragma unroll
                        for (int a = 0; a < D - 1; ++a) twc[s][a] = twn[s][a];
                    }
                    go = go_n;
                }
            }  // o1
        }      // o0
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!valid[s]) continue;
            int label;
            if (C == 1) {
                label = best_j[s];
            } else if (C == 2) {
                label = best_uo[s] + P->m[0] * best_j[s];
            } else {
                const int j1 = best_uo[s] % P->m[1], j0 = best_uo[s] / P->m[1];
                label = j0 + P->m[0] * (j1 + P->m[1] * best_j[s]);
            }
            const int in_plane = ls[s] % inner_sz, pl = ls[s] / inner_sz;
            Jout[in_plane + inner_sz * (pl + P->halo_lo)] = best[s];
            if (idx_out) st_idx(idx_out, ls[s], label + P->index_base, P->idx_bytes);
        }
    }
}

}  // namespace hjb
