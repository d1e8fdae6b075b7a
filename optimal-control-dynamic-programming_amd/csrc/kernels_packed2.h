// kernels_packed2.h - variant 4: the throughput kernel of the canonical spacecraft shape, one state per lane, packed fp32.
// Like variant 2 (kernels_packed.h) it runs the innermost control loop with no cell test, no search and no load on the vector
// pipe, but its per-lane state is half as large (more waves per SIMD; C2: 16,100 waves instead of 8,050).
//   * the sweep keeps MINIMA only (3 packed operations + one v_min3 per control pair and step); the first (o0, o1) step that
//     attains the state's minimum is tracked per step, and that step's controls are re-evaluated once per state, in order, to
//     find the first one that reproduces it - the label (first-minimum rule, bit for bit);
//   * in the hierarchical modes two (o0, o1) steps run per trip with the two STEPS in the halves of every packed instruction;
//     the one-step loop (anything irregular, the last step of an odd count) puts two consecutive CONTROLS there instead;
//   * the window modes (MODE 2, 3, 5, 6) contract the state-only axes once per state into an LDS window.
// Arithmetic per backup is the canonical order throughout: results are bit-identical to the oracle and to every other variant.
#pragma once
#include <type_traits>
#include "hjbdp_dev.h"
#include "kernels_nested.h"
#include "kernels_packed.h"

namespace hjb {

// One 8-byte LDS read that stays one.  Left to itself the compiler merges two of them into ds_read2(st64)_b64, which runs at
// half the LDS rate (MI355X_MICROARCH.md, LDS: ds_read_b64 256 B/clk, ds_read2_b64 128 B/clk); the pair loop issues one per
// lane and one broadcast per control pair and was LDS-bound on them (C2 1.60 -> 1.55 ms, 24^6 97 -> 95 ms per stage).
__device__ __forceinline__ f2 lds_f2(const f2 *p) {
    return *(const volatile __attribute__((address_space(3))) f2 *)p;
}

// The 2^NP corners of the NP leading axes at element offset `off`, and their contraction (lerp order: axis 0 first).  Two
// functions so that a caller can put the gathers of several window entries in flight before the first lerp waits for one.
template <typename TJ, int NP>
__device__ __forceinline__ void gather_corners(const TJ *__restrict__ Jn, int off, const int (&js)[NP + 3], float (&v)[1 << NP]) {
#pragma unroll
    for (int c = 0; c < (1 << NP); ++c) {
        int o = off;
#pragma unroll
        for (int a = 0; a < NP; ++a) o += ((c >> a) & 1) ? js[a] : 0;
        v[c] = (float)Jn[o];
    }
}

template <int NP>
__device__ __forceinline__ float contract_corners(float (&v)[1 << NP], const float (&tw)[NP + 2]) {
#pragma unroll
    for (int a = 0; a < NP; ++a) {
#pragma unroll
        for (int jj = 0; jj < (1 << (NP - 1 - a)); ++jj) v[jj] = __builtin_fmaf(tw[a], v[2 * jj + 1] - v[2 * jj], v[2 * jj]);
    }
    return v[0];
}

// ---- HJB_MODEL_QUAT_EULER321 (hjbdp.h): next (yaw, pitch, roll) of one state ---------------------------------
// attitude-control/Solver_attitude.m:449-489 in single precision, operation by operation.  atan2/asin are fixed
// polynomial forms made of +,-,*,/ and sqrt (all correctly rounded on gfx950 - NB __builtin_sqrtf, not __fsqrt_rn,
// which lowers to the bare 1-ulp v_sqrt_f32 - and contraction off), so the result is
// reproducible bit for bit on any IEEE machine - the checker (oracle/hjb_oracle.c) restates the same forms.
__device__ __forceinline__ float canon_atan2f(float y, float x) {
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const bool swap = ay > ax;
    const float num = swap ? ax : ay, den = swap ? ay : ax;
    float r = den == 0.0f ? 0.0f : __fdiv_rn(num, den);
    float off = 0.0f;
    if (r > 0.4142135623730950f) {
        r = __fdiv_rn(r - 1.0f, r + 1.0f);
        off = 0.78539816339744831f;
    }
    const float z = r * r;
    float pz = 8.05374449538e-2f;
    pz = pz * z - 1.38776856032e-1f;
    pz = pz * z + 1.99777106478e-1f;
    pz = pz * z - 3.33329491539e-1f;
    float a = off + (pz * z * r + r);
    if (swap) a = 1.57079632679489662f - a;
    if (x < 0.0f) a = 3.14159265358979324f - a;
    return y < 0.0f ? -a : a;
}

__device__ __forceinline__ float canon_asinf(float x) {
    const float a = __builtin_fabsf(x);
    const bool big = a > 0.5f;
    float z, r;
    if (big) {
        z = 0.5f * (1.0f - a);
        r = __builtin_sqrtf(z);
    } else {
        r = a;
        z = a * a;
    }
    float pz = 4.2163199048e-2f;
    pz = pz * z + 2.4181311049e-2f;
    pz = pz * z + 4.5470025998e-2f;
    pz = pz * z + 7.4953002686e-2f;
    pz = pz * z + 1.6666752422e-1f;
    float v = pz * z * r + r;
    if (big) v = 1.57079632679489662f - (v + v);
    return x < 0.0f ? -v : v;
}

__device__ __forceinline__ void model_quat_next(const DParams *__restrict__ P, const int *si, float w1, float w2, float w3,
                                                float (&out)[3]) {
    const int ti = si[0] + P->axis[0].n * (si[1] + P->axis[1].n * si[2]);
    const float q1 = as_global<float>(P->model_tab[0])[ti], q2 = as_global<float>(P->model_tab[1])[ti];
    const float q3 = as_global<float>(P->model_tab[2])[ti], q7 = as_global<float>(P->model_tab[3])[ti];
    const float h = P->model_h, half = 0.5f;
    float x4 = q1 + h * (half * ((w3 * q2 - w2 * q3) + w1 * q7));       // :449-452
    float x5 = q2 + h * (half * ((-w3 * q1 + w1 * q3) + w2 * q7));      // :454-457
    float x6 = q3 + h * (half * ((w2 * q1 - w1 * q2) + w3 * q7));       // :459-462
    float x7 = q7 + h * (half * ((-w1 * q1 - w2 * q2) - w3 * q3));      // :465-467
    const float nrm = __builtin_sqrtf(((x4 * x4 + x5 * x5) + x6 * x6) + x7 * x7);   // :477
    x4 = __fdiv_rn(x4, nrm); x5 = __fdiv_rn(x5, nrm); x6 = __fdiv_rn(x6, nrm); x7 = __fdiv_rn(x7, nrm);   // :480-483
    out[0] = canon_atan2f(2.0f * (x6 * x5 + x7 * x4), ((x7 * x7 + x6 * x6) - x5 * x5) - x4 * x4);   // :485-486
    out[1] = canon_asinf(-2.0f * (x6 * x4 - x7 * x5));                                              // :487
    out[2] = canon_atan2f(2.0f * (x5 * x4 + x7 * x6), ((x7 * x7 - x6 * x6) - x5 * x5) + x4 * x4);   // :488-489
}

// find_cell (kernels_generic.h) on a global-address-space knot vector: the same exact search
__device__ __forceinline__ int find_cell_g(gptr<float> k, int n, float q, int uniform, float x0, float inv_h) {
    int i;
    if (uniform) {
        float f = (q - x0) * inv_h;
        const float hi = (float)(n - 2);
        f = f > 0.f ? f : 0.f;
        f = f < hi ? f : hi;
        i = (int)f;
        while (i > 0 && q < k[i]) --i;
        while (i < n - 2 && q >= k[i + 1]) ++i;
    } else {
        int lo = 0, hi = n - 1;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (k[mid] <= q) lo = mid; else hi = mid;
        }
        i = lo;
    }
    return i;
}

// Depth-first contraction of axes 0..A-1 at element offset `off` (axis 0 lerped first, like contract<>): the same
// tree as the breadth-first form, hence the same bits, with O(A) live registers instead of 2^A.  For rare paths.
template <typename TJ, int A, int DD>
__device__ __forceinline__ float contract_df(const TJ *__restrict__ Jn, int off, const int (&js)[DD], const float *tw) {
    if constexpr (A == 0) {
        return (float)Jn[off];
    } else if constexpr (A <= 3) {
        const float v0 = contract_df<TJ, A - 1, DD>(Jn, off, js, tw);
        const float v1 = contract_df<TJ, A - 1, DD>(Jn, off + js[A - 1], js, tw);
        return __builtin_fmaf(tw[A - 1], v1 - v0, v0);
    } else {                       // upper levels as real loops: 8 loads in flight, small code
        float v0 = 0.f, v1 = 0.f;
#pragma unroll 1
        for (int hh = 0; hh < 2; ++hh) {
            const float x = contract_df<TJ, A - 1, DD>(Jn, off + hh * js[A - 1], js, tw);
            if (hh == 0) v0 = x; else v1 = x;
        }
        return __builtin_fmaf(tw[A - 1], v1 - v0, v0);
    }
}

// MODE (chosen on the host from the axis levels; "level" = the outermost control loop an axis' cell depends on):
//   0  plain: 2 x 2^D corner gathers + full contraction per (o0,o1) step
//   1  D == 3, axis 0 level 0, axis 1 level 1 (the C2 shape): axis 0 contracted once per o0 step
//   4  mode 1 with axis 0's (cell, t) formed per o0 step from q in registers instead of read from a table (the host picks it
//      when axis 0's next value is state-only terms + ONE term over control dim 0: no 8-byte-per-(state, o0) table at all)
//   2  D >= 4, axes 0..D-4 state-only, axis D-3 level 0, axis D-2 level 1 (the attitude model with the angle axes
//      first): the state-only axes are contracted ONCE PER STATE over the 3 x 3 x 4 window of (axis D-3 rows,
//      axis D-2 rows, last-axis planes) the whole control sweep can touch; an o0 step is then 24 selects + 12
//      lerps and an o1 step 8 selects + 4 lerps, whatever D is.
//   3  mode 2 with D == 6 and the three leading axes driven by HJB_MODEL_QUAT_EULER321 (their next value is
//      computed per state instead of read from nS-sized tables) and 64-bit state indexing: C3, 51^6 states.
// Same lerp order (axis 0 first ... last axis last) -> same bits in every mode.
#ifndef HJB_K3_M1_WAVES
#define HJB_K3_M1_WAVES 5
#endif
template <typename TJ, int D, int MODE>
// The C2 modes are held to 96 VGPRs = five waves per SIMD (ten values spilled; 1.40 -> 1.35 ms per stage on C2; six waves
// = 80 VGPRs spill 44 and run 1.7x slower); the window modes sit at three workgroups per CU by LDS whatever the registers.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MODE == 4 || MODE == 1) ? HJB_K3_M1_WAVES : ((MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6) ? 4 : 1))))
k_backup_packed2(const DParams *__restrict__ P, const DNested *__restrict__ N, const TJ *__restrict__ Jn,
                 TJ *__restrict__ Jout, void *__restrict__ idx_out) {
    constexpr int DM = D > 1 ? D - 1 : 1;
    constexpr bool INL0 = MODE == 4;          // mode 1 with axis 0's (cell, t) formed in the kernel (no table: see below)
    constexpr bool M1 = MODE == 1 || MODE == 4;
    constexpr bool HIER = MODE != 0;
    constexpr int AX_A = D >= 3 ? D - 3 : 0, AX_B = D >= 2 ? D - 2 : 0;   // the level-0 / level-1 axes of modes 1, 2
    constexpr bool W3P = MODE == 5 || MODE == 6;            // modes 2 / 3 with the three-plane window (see kWin)
    constexpr bool PRE = MODE == 2 || MODE == 3 || W3P, QMODEL = MODE == 3 || MODE == 6;
    constexpr int NP = PRE ? D - 3 : 0;
    using sidx_t = typename std::conditional<QMODEL, int64_t, int>::type;   // linear state index
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // The C2 modes (five waves per SIMD = 96 registers) park per-state values that are read once per o0 step in LDS, by hand:
    // left to the compiler they went to scratch (round 4: 67 MB of spill traffic per C2 stage).  No forwarding of
    // the stored value, which would keep it in its register (a compiler barrier behind the store: a volatile access would turn
    // into a flat store with system scope).
    __shared__ int s_park[M1 ? 2 : 1][M1 ? 256 : 1];
    const DAxis &axl = P->axis[D - 1];
    const int nl = axl.n;
    const int m_in = N->m_in;
    const int npairs = (m_in + 1) >> 1;
    // LDS: {t_2p, t_2p+1} per lane [npairs+1][256] float2 | {r_2p, r_2p+1} [npairs+1] | b[m_in] |
    //      knots, rdx of the last axis | control-only cost tables
    // mode 2 only: the per-state window W[(ra*3+rb)*4+q][lane] in front of everything else
    // W3P (any D >= 4): the host has checked that the inner control moves the last axis by less than one cell per step, so the second cell
    // a sweep enters is a neighbour of the first and the window needs 3 last-axis planes, not 4: 27 entries - with no padding
    // row in the weights 40 KB of LDS per workgroup on the 11-torque attitude grids, i.e. FOUR workgroups per CU instead of three
    constexpr int kPairsUnrolled = M1 ? 11 : 6;            // the pair-row count the two-step trips' sweep is written out for: the window modes' 11 / 12
                                                           // inner controls (6-D attitude grids, C3); modes 1 / 4: 21 (C2's 21^3 controls)
    constexpr int kWin = PRE ? (W3P ? 27 : 36) : 0;
    constexpr int kWq = W3P ? 3 : 4;                        // window planes
    float *my_w = reinterpret_cast<float *>(smem_raw) + threadIdx.x;
    f2 *s_t = reinterpret_cast<f2 *>(smem_raw + (size_t)kWin * 256 * sizeof(float));
    const int t_rows = W3P ? npairs : npairs + 1;           // W3P: no padding row (the read-ahead is clamped instead)
    f2 *s_r2 = s_t + (size_t)t_rows * 256;
    float *s_b = reinterpret_cast<float *>(s_r2 + (npairs + 1));
    float *s_k = s_b + m_in;
    float *s_r = s_k + nl;
    float *s_ot = s_r + nl;
    for (int i = threadIdx.x; i < nl; i += blockDim.x) {
        s_k[i] = static_cast<const float *>(axl.knots)[i];
        s_r[i] = static_cast<const float *>(axl.rdx)[i];
    }
    {
        const DInnerTerm &tb = N->in[0];
        const DInnerTerm &tr = N->in[kMaxInAx];
        if (tb.lds_slot >= 0)   // control-only inner table; a state-dependent one is read from global per state
            for (int i = threadIdx.x; i < m_in; i += blockDim.x) s_b[i] = static_cast<const float *>(tb.data)[i * tb.stride_in];
        for (int p = threadIdx.x; p <= npairs; p += blockDim.x) {
            f2 x = {INFINITY, INFINITY};          // padding controls: infinite cost, never selected
            if (p == npairs) x = (f2){-0.0f, -0.0f};   // the row after the last pair (read ahead, never used as a control) doubles as the
                                                      // "no level-1 cost term" slot of the two-step trips: g + (-0) == g bit for bit
            if (2 * p < m_in) x.x = static_cast<const float *>(tr.data)[(2 * p) * tr.stride_in];
            if (2 * p + 1 < m_in) x.y = static_cast<const float *>(tr.data)[(2 * p + 1) * tr.stride_in];
            s_r2[p] = x;
        }
        if constexpr (!W3P) s_t[(size_t)npairs * 256 + threadIdx.x] = (f2){0.f, 0.f};
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
    const sidx_t n_owned = (sidx_t)P->n_owned;
    const int l_uniform = axl.uniform;
    const float l_x0 = (float)axl.x0, l_invh = (float)axl.inv_h;
    const int plane0 = P->plane0, nplanes = P->nplanes;
    const int m_o0 = N->m_o0, m_o1 = N->m_o1;
    int js[D];
#pragma unroll
    for (int a = 0; a < D; ++a) js[a] = a == 0 ? 1 : (int)P->jstride[a];      // (axis 0 is contiguous by construction: a literal 1 lets a corner's
                                                                               //  axis-0 neighbour be an immediate offset of the same address)
    const int inner_sz = (int)P->inner;
    f2 *my_t = s_t + threadIdx.x;
    // the weights row read ahead of pair p (beyond the last pair: the padding row, or - without one - the last row again)
    auto t_ahead = [&](int p) __attribute__((always_inline)) -> int {
        if constexpr (W3P) return p + 1 < npairs ? p + 1 : p;
        else return p + 1;
    };
    gptr<i2v> atab[DM];
    int a_c0[DM], a_c1[DM], a_lvl_rt[DM];
#pragma unroll
    for (int a = 0; a < D - 1; ++a) {
        atab[a] = as_global<i2v>(N->at[a].tab);
        a_c0[a] = N->at[a].c0;
        a_c1[a] = N->at[a].c1;
        a_lvl_rt[a] = N->at[a].level;
    }
    // an entry of axis a's table by ELEMENT index: a 32-bit byte offset against the table's scalar base (tables are < 4 GiB; a
    // per-lane 64-bit pointer held across the control loops was the last register pair the C2 modes kept in scratch)
    auto at_ld = [&](int a, int elem) __attribute__((always_inline)) -> i2v {
        return *reinterpret_cast<gptr<i2v>>(reinterpret_cast<gptr<char>>(atab[a]) + (uint32_t)elem * 8u);
    };
    // axis a's per-state table offset: from LDS in the C2 modes (see s_park), else the register
    auto aoff_of = [&](int a, const int (&aoff_r)[DM]) __attribute__((always_inline)) -> int {
        if constexpr (M1) return a < 2 ? s_park[a][threadIdx.x] : aoff_r[a];
        else return aoff_r[a];
    };
    // the level of axis a's table: in modes 2 and 3 the host has checked the pattern (axis D-3 level 0, axis D-2 level 1, every
    // axis before them state-only), so it is a compile-time constant of the unrolled axis loops (24^6: 166 -> 118 scalar
    // registers spilled to vector lanes, 70 fewer lane reads per o0 step); the other modes read it
    auto a_lvl = [&](int a) __attribute__((always_inline)) -> int {
        if constexpr (PRE) return a == AX_A ? 0 : (a == AX_B ? 1 : -1);
        else return a_lvl_rt[a];
    };
    const bool cl0_present = N->ot[CL0].present, cl1_present = N->ot[CL1].present;
    const bool cl0_first = N->ot[CL0].first, cl1_first = N->ot[CL1].first;
    const int cl_off[2] = {N->ot[CL0].lds_off, N->ot[CL1].lds_off};
    const int cl_c0[2] = {N->ot[CL0].c0, N->ot[CL1].c0}, cl_c1[2] = {N->ot[CL0].c1, N->ot[CL1].c1};
    // last axis' inner term: control-only (LDS) or state-dependent (global, offset boff + j * stride)
    // Mode 1 (the C2 shape): when axis 0's next value is (state-only terms) + ONE term over control dim 0, its (cell, t) is
    // formed per o0 step from q in registers - the same ordered sum, the same exact search, the same weight as the table
    // entry would hold - and no table is built: C2's 173 MB of axis-0 entries (8 bytes per state and o0 step, re-read every
    // stage: 9x the algorithmic HBM bytes) disappear.  N->at[0].tab == nullptr says so.
    // axis 0's (cell, t) at control o0 from the state part qs of its next value: read from P where it is needed (once per o0
    // step), not hoisted - the kernel sits at the register budget of four waves per SIMD
    auto axis0_entry = [&](float qs, int o0, int &c_out, float &t_out) __attribute__((always_inline)) {
        const DAxis &a0 = P->axis[0];
        const int npre0 = a0.n_prefix;
        const float bq = as_global<float>(a0.t[npre0].data)[o0];
        const float q0 = npre0 == 0 ? bq : qs + bq;
        gptr<float> kk0 = as_global<float>(a0.knots);
        const int c0 = find_cell_g(kk0, a0.n, q0, a0.uniform, (float)a0.x0, (float)a0.inv_h);
        c_out = c0;
        t_out = (q0 - kk0[c0]) * as_global<float>(a0.rdx)[c0];
    };
    const bool b_pure = N->in[0].lds_slot >= 0;
    gptr<float> b_data = as_global<float>(N->in[0].data);
    const int b_stride = N->in[0].stride_in;

    // Window modes, the order in which the 256-state chunks are visited.  (1) Workgroups are handed to the eight XCDs
    // round-robin, so neighbouring chunks would sit under eight different L2s: XCD x takes the x-th CONTIGUOUS eighth of every
    // grid-sized span of the visiting order.  (2) In state order the chunks of one point of the level axes (all values of the
    // state-only axes: `ac` chunks) come first, so the chunks whose windows overlap most - the neighbours along the level-0
    // and level-1 axes - are ac, ac * n apart and never resident together.  The visiting order is the transpose: for one
    // chunk of the state-only axes, all points of the other axes in turn (v -> chunk (v mod nw) * ac + v / nw over the
    // ac x nw rectangle that covers the chunks; positions past the last chunk are skipped).
    unsigned int first_v = blockIdx.x, n_v = 0, ac = 1, nw = 1;
    const unsigned int n_chunks = (unsigned int)((n_owned + 255) / 256);
    if constexpr (PRE) {
        if ((gridDim.x & 7u) == 0u) first_v = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        unsigned int inner = 1;
#pragma unroll
        for (int a = 0; a < NP; ++a) inner *= (unsigned int)P->n[a];
        ac = inner >= 256u ? inner / 256u : 1u;
        // option "chunk_order" = 1: state order instead (each XCD a contiguous run of chunks = neighbouring angle chunks of
        // ONE point of the level axes: what a grid whose angle block outgrows the L2 wants - C3: 51^3 x 27 window slices =
        // 14 MB per point - where the transposed order re-fetches the slices for every chunk)
        if (N->chunk_order == 1) ac = 1u;
        nw = (n_chunks + ac - 1) / ac;
        n_v = ac * nw;
    } else {
        n_v = n_chunks;
        // the same locality rule without a window: workgroup b runs on XCD b % 8, so with chunk = b every XCD's L2 fetched nearly
        // the whole J (C2: 24 MB through the fabric for a 4 MB array).  XCD x takes the x-th contiguous share of every grid-sized span
        // (its workgroups b = x, x + 8, ...: G / 8 of them, one more for x < G % 8)
        const unsigned int G = gridDim.x, x = blockIdx.x & 7u, q = G >> 3, r = G & 7u;
        first_v = x * q + (x < r ? x : r) + (blockIdx.x >> 3);
    }
    for (unsigned int v = first_v; v < n_v; v += gridDim.x) {
        unsigned int chunk = v;
        if constexpr (PRE) {
            const unsigned int va = v / nw;
            chunk = (v - va * nw) * ac + va;
            if (chunk >= n_chunks) continue;
        }
        const sidx_t blk = (sidx_t)chunk * 256;
        sidx_t ls = blk + threadIdx.x;
        const bool valid = ls < n_owned;
        if (!valid) ls = n_owned - 1;         // harmless duplicate work, store skipped

        float ql, gpre, qs0 = 0.f;            // qs0: state-only part of axis 0's next value (inl0)
        int boff = 0;                         // state part of the inner term's table offset
        int aoff[DM], coff[2], cell[DM];
        float tw[DM];
        int lc0, lc1;
        unsigned int cm = 0u;                 // bit j: the last-axis cell changes at control j
        {
            int si[D];
            if constexpr (QMODEL) {               // 64-bit state index (C3)
                sidx_t r = ls;
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    int na = P->n[a];
                    si[a] = (int)(r % na);
                    r /= na;
                }
            } else {
                // 32-bit index: division by the grid sizes with the host's multipliers (DNested::div_m): the compiler's expansion of
                // `r / n[a]` keeps a reciprocal per divisor in a vector register for the whole kernel (uniform values computed on the
                // vector unit) - five of the eleven registers the C2 modes spilled at five waves per SIMD
                uint32_t r = (uint32_t)ls;
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const uint32_t q = udiv_gm(r, N->div_m[a], N->div_s[a]);
                    si[a] = (int)(r - q * (uint32_t)P->n[a]);
                    r = q;
                }
            }
            const int last_local = si[D - 1];
            si[D - 1] += P->slab_begin;
            float q = 0.f;
            for (int k = 0; k < axl.n_prefix; ++k) {
                float x = term_value32<D>(axl.t[k], si);
                q = (k == 0) ? x : q + x;
            }
            ql = q;
            if (!b_pure) {
                const DTerm &bt = axl.t[N->ax_kin];
#pragma unroll
                for (int a = 0; a < D; ++a) boff += bt.stride[a] * si[a];
            }
            float g = 0.f;
            for (int k = 0; k < P->n_cost_prefix; ++k) {
                float x = term_value32<D>(P->cost[k], si);
                g = (k == 0) ? x : g + x;
            }
            gpre = g;
            if constexpr (INL0) {
                float q0 = 0.f;
                for (int k = 0; k < P->axis[0].n_prefix; ++k) {
                    const float x = term_value32<D>(P->axis[0].t[k], si);
                    q0 = (k == 0) ? x : q0 + x;
                }
                qs0 = q0;
            }
#pragma unroll
            for (int a = 0; a < D - 1; ++a) {
                int off = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) off += N->at[a].sstride[d] * (d == D - 1 ? last_local : si[d]);
                aoff[a] = off;
                if constexpr (M1) {
                    if (a < 2) {
                        s_park[a][threadIdx.x] = off;
                        asm volatile("" ::: "memory");         // no forwarding of the stored value to the reads below
                    }
                }
                if (a_lvl(a) < 0 && !(QMODEL && a < 3)) {
                    const i2v e = at_ld(a, off);
                    cell[a] = e.x;
                    tw[a] = __int_as_float(e.y);
                }
            }
            if constexpr (QMODEL) {
                float qn[3];
                model_quat_next(P, si, as_global<float>(P->axis[3].knots)[si[3]],
                                as_global<float>(P->axis[4].knots)[si[4]], s_k[si[5]], qn);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const DAxis &ax = P->axis[a];
                    gptr<float> kk = as_global<float>(ax.knots);
                    const int c = find_cell_g(kk, ax.n, qn[a], ax.uniform, (float)ax.x0, (float)ax.inv_h);
                    cell[a] = c;
                    tw[a] = (qn[a] - kk[c]) * as_global<float>(ax.rdx)[c];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const auto &t = N->ot[CL0 + i];
                int off = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) off += t.sstride[d] * si[d];
                coff[i] = off;
            }
        }
        // ---- once per state: inner weights t_j (pairs) and cell-crossing bits --------
        {
            CellTrack<float> tl;
            track_reset(tl);
            f2 t = {0.f, 0.f};
            for (int j = 0; j < m_in; ++j) {
                const float q = ql + (b_pure ? s_b[j] : b_data[boff + j * b_stride]);
                const bool ch = track_update<float>(tl, s_k, s_r, nl, q, l_uniform, l_x0, l_invh);
                if (j == 0) {
                    lc0 = lc1 = tl.cell - plane0;
                } else if (ch) {
                    if (cm == 0u) lc1 = tl.cell - plane0;
                    cm |= 1u << j;
                }
                const float tj = (q - tl.kc) * tl.rc;
                if (j & 1) { t.y = tj; my_t[(j >> 1) * 256] = t; }
                else { t.x = tj; t.y = 0.f; if (j == m_in - 1) my_t[(j >> 1) * 256] = t; }
            }
        }
        unsigned int U = 0u;
        for (int j = 1; j < m_in; ++j)
            if (__ballot((cm >> j) & 1u)) U |= 1u << j;
        unsigned int PU = 0u;                 // bit p: some lane changes cell at control 2p or 2p+1
        for (int p = 0; p < npairs; ++p)
            if ((U >> (2 * p)) & 3u) PU |= 1u << p;
        // UA bit j: EVERY lane of the wave makes its first cell change at control j (e.g. C2: the whole wave
        // crosses where the control changes sign) - then the switch to the prefetched second cell is scalar
        // control flow, no per-lane test.  UX bit j: some lane changes cell at j in any other way.
        unsigned int UA = 0u;
        {
            const unsigned int cm_first = cm & (0u - cm);
            const unsigned long long all = __ballot(1);
            for (int j = 1; j < m_in; ++j)
                if ((U >> j) & 1u)
                    if (__ballot((cm_first >> j) & 1u) == all) UA |= 1u << j;
        }
        const unsigned int UX = U & ~UA;
        // UX == 0: every lane's cm is the same single bit (or 0) - the control at which the wave enters the second cell
        const int jc = U ? __builtin_ctz(U) : m_in;
        if (lc0 < 0 || lc0 + 1 >= nplanes) { *P->status = 1; lc0 = lc0 < 0 ? 0 : nplanes - 2; }
        if (lc1 < 0 || lc1 + 1 >= nplanes) { *P->status = 1; lc1 = lc1 < 0 ? 0 : nplanes - 2; }
        // W3P: window plane of each cell's lower corner; a state whose second cell is NOT a neighbour of the first (the host's
        // check makes that impossible) is served by the synchronous gathers like any step outside the window
        const int pl0 = lc0 < lc1 ? lc0 : lc1;
        const int qa = lc0 - pl0, qb = lc1 - pl0;
        const bool far_cells = W3P && (qa > 1 || qb > 1);
        // smallest cell over `cnt` table entries `step` apart: four independent loads in flight per trip
        auto min_cell = [&](int a, int e0, int cnt, int step) {
            int cm = 0x7fffffff, o = 0;
            for (; o + 4 <= cnt; o += 4) {
                const int x0 = at_ld(a, e0 + o * step).x, x1 = at_ld(a, e0 + (o + 1) * step).x, x2 = at_ld(a, e0 + (o + 2) * step).x,
                          x3 = at_ld(a, e0 + (o + 3) * step).x;
                const int lo01 = x0 < x1 ? x0 : x1, lo23 = x2 < x3 ? x2 : x3;
                const int lo = lo01 < lo23 ? lo01 : lo23;
                cm = lo < cm ? lo : cm;
            }
            for (; o < cnt; ++o) {
                const int x = at_ld(a, e0 + o * step).x;
                cm = x < cm ? x : cm;
            }
            return cm;
        };
        // modes 1-3: does the level-1 axis' table depend on o0 as well?  (wave-uniform; usually not)
        const bool b_o0dep = HIER && a_c0[AX_B] != 0;
        int cAmin = 0, cBmin = 0;
        if constexpr (HIER) {
            if (!b_o0dep) cBmin = min_cell(AX_B, aoff_of(AX_B, aoff), m_o1, a_c1[AX_B]);
        }
        if constexpr (PRE) {
            const int ca = min_cell(AX_A, aoff_of(AX_A, aoff), m_o0, a_c0[AX_A]);
            if (b_o0dep) {
                int cbm = 0x7fffffff;
                for (int o0 = 0; o0 < m_o0; ++o0) {
                    const int c2 = min_cell(AX_B, aoff_of(AX_B, aoff) + o0 * a_c0[AX_B], m_o1, a_c1[AX_B]);
                    cbm = c2 < cbm ? c2 : cbm;
                }
                cBmin = cbm;
            }
            const int cb = cBmin;
            cAmin = ca;
            const int nA = P->axis[AX_A].n, nB = P->axis[AX_B].n;
            // window planes: the two cells' four planes, or (W3P, neighbouring cells) the three planes from the lower cell up
            int planes[kWq];
            if constexpr (W3P) {
                planes[0] = pl0;
                planes[1] = pl0 + 1;
                planes[2] = pl0 + 2 < nplanes ? pl0 + 2 : nplanes - 1;      // not read from when the sweep stays in one cell
            } else {
                planes[0] = lc0; planes[1] = lc0 + 1; planes[2] = lc1; planes[3] = lc1 + 1;
            }
            int pbase = 0;
#pragma unroll
            for (int a = 0; a < NP; ++a) pbase += js[a] * cell[a];
#pragma unroll 1
            for (int ra = 0; ra < 3; ++ra) {
                const int rowA = ca + ra < nA ? ca + ra : nA - 1;
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    const int rowB = cb + rb < nB ? cb + rb : nB - 1;
                    const int off = pbase + js[AX_A] * rowA + js[AX_B] * rowB;      // < one plane: fits 32 bits
                    // the planes' corners first (4 x 2^NP gathers in flight), then their lerps: one wait per (ra, rb)
                    // instead of one per window entry (24^6: 36 -> 9 round trips to L2 / HBM per state)
                    float v[kWq][1 << NP];
#pragma unroll
                    for (int q = 0; q < kWq; ++q) gather_corners<TJ, NP>(Jn + (int64_t)js[D - 1] * planes[q], off, js, v[q]);
#pragma unroll
                    for (int q = 0; q < kWq; ++q) my_w[((ra * 3 + rb) * kWq + q) * 256] = contract_corners<NP>(v[q], tw);
                }
            }
        }
        float best = 0.f;
        int best_uo = 0;

        // (E0, dE) of one last-axis cell at element offset `off`: the rare synchronous path
        auto cell_pair = [&](int base, int lc, const float (&twc)[DM], float &e0, float &de) {
            if constexpr (PRE) {
                const TJ *Jp = Jn + (int64_t)js[D - 1] * lc;
                e0 = contract_df<TJ, D - 1, D>(Jp, base, js, twc);
                de = contract_df<TJ, D - 1, D>(Jp + js[D - 1], base, js, twc) - e0;
            } else {
                float v[1 << D];
                load_corners<D>(Jn, base + js[D - 1] * lc, js, v);
                contract<D>(v, twc, e0, de);
            }
        };
        auto cterm = [&](int slot, int o0, int o1) -> float {
            if constexpr (HIER) {      // modes 1-3 are only chosen when the level cost terms are control-only: LDS
                const int i = slot - CL0;
                return s_ot[cl_off[i] + o0 * cl_c0[i] + o1 * cl_c1[i]];
            } else {
                const auto &t = N->ot[slot];
                if (t.lds_off >= 0) return s_ot[t.lds_off + o0 * t.c0 + o1 * t.c1];
                return as_global<float>(t.data)[coff[slot - CL0] + o0 * t.c0 + o1 * t.c1];
            }
        };

        i2v l0_nx = {0, 0};                   // window modes: the level-0 axis' entry, fetched one o0 step ahead
        if constexpr (PRE) l0_nx = at_ld(AX_A, aoff_of(AX_A, aoff));
#ifndef HJB_K3_PROBE
#define HJB_K3_PROBE 0
#endif
        for (int o0 = 0; o0 < (HJB_K3_PROBE == 2 ? 0 : m_o0); ++o0) {
            // ---- level 0 ------------------------------------------------------------
            if constexpr (PRE) {
                cell[AX_A] = l0_nx.x;
                tw[AX_A] = __int_as_float(l0_nx.y);
                if (o0 + 1 < m_o0) l0_nx = at_ld(AX_A, aoff_of(AX_A, aoff) + (o0 + 1) * a_c0[AX_A]);
            } else {
#pragma unroll
                for (int a = 0; a < D - 1; ++a) {
                    if (a_lvl(a) == 0) {
                        if (INL0 && a == 0) {
                            axis0_entry(qs0, o0, cell[a], tw[a]);
                        } else {
                            const i2v e = at_ld(a, aoff_of(a, aoff) + o0 * a_c0[a]);
                            cell[a] = e.x;
                            tw[a] = __int_as_float(e.y);
                        }
                    }
                }
            }
            const int uo0 = o0 * m_o1;                 // (o0, o1) step number = uo0 + o1: scalar, never a vector register
            float go0 = gpre;
            if (cl0_present) {
                const float x = cterm(CL0, o0, 0);
                go0 = cl0_first ? x : go0 + x;
            }
            auto prepare = [&](int o1, int &base, float (&twc)[DM], float &go) {
                int b = 0;
#pragma unroll
                for (int a = 0; a < D - 1; ++a) {
                    if (a_lvl(a) == 1) {
                        const i2v e = at_ld(a, aoff_of(a, aoff) + o0 * a_c0[a] + o1 * a_c1[a]);
                        cell[a] = e.x;
                        tw[a] = __int_as_float(e.y);
                    }
                    b += js[a] * cell[a];
                    twc[a] = tw[a];
                }
                base = b;
                float g = go0;
                if (cl1_present) {
                    const float x = cterm(CL1, o0, o1);
                    g = cl1_first ? x : g + x;
                }
                go = g;
            };
            int base, base_n;
            float twc[DM], twn[DM];
            float go, go_n;
            float G[1 << D], H[1 << D];   // prefetched corners (unused, hence eliminated, when HIER)
            // D == 3 with axis 0 resolved per o0 step and axis 1 per (o0,o1) step (the C2 shape): contract
            // axis 0 ONCE per o0 step for the <= 3 axis-1 rows and the 4 last-axis planes the whole o1 sweep
            // can touch; an o1 step is then 8 selects + 4 lerps instead of 16 loads + 13 lerps.  Same lerp
            // order (axis 0, then 1, then the last axis) -> same bits.
            float F[HIER ? 3 : 1][4];
            int c1min = 0;
            bool slow_a = false;                                     // mode 2: axis D-3 left its 2-cell window
            if constexpr (PRE) {
                c1min = cBmin;
                const int r = cell[AX_A] - cAmin;
                slow_a = !(r == 0 || r == 1) || far_cells;
                const float *w0 = my_w + (r == 1 ? 3 * kWq * 256 : 0);     // rows {r, r+1} of the window
                const float ta = tw[AX_A];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // window plane of F's plane q = {cell 0 lower, upper, cell 1 lower, upper}
                        const int wq = W3P ? (q < 2 ? qa : qb) + (q & 1) : q;
                        const float f0 = w0[(rb * kWq + wq) * 256];
                        const float f1 = w0[(3 * kWq + rb * kWq + wq) * 256];
                        F[rb][q] = __builtin_fmaf(ta, f1 - f0, f0);
                    }
            } else if constexpr (M1) {
                const int cmin = b_o0dep ? min_cell(1, aoff_of(1, aoff) + o0 * a_c0[1], m_o1, a_c1[1]) : cBmin;
                c1min = cmin;
                const int n1 = P->axis[1].n;
                const int rows[3] = {cmin, cmin + 1, cmin + 2 < n1 ? cmin + 2 : n1 - 1};
                const int planes[4] = {lc0, lc0 + 1, lc1, lc1 + 1};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v2[2];
                        load_pair(Jn, cell[0] + js[1] * rows[r] + js[2] * planes[q], v2);
                        F[r][q] = __builtin_fmaf(tw[0], v2[1] - v2[0], v2[0]);
                    }
                }
            } else {
                prepare(0, base, twc, go);
                load_corners<D>(Jn, base + js[D - 1] * lc0, js, G);
                load_corners<D>(Jn, base + js[D - 1] * lc1, js, H);
            }
            // modes 1-3: the level-1 axis' (cell, t) entry and the level-1 cost term are fetched ONE STEP AHEAD
            // (a dependent global load per o1 step would otherwise stall every step)
            // the level-1 axis' entries of this o0 step: a 32-bit BYTE offset against the table's scalar base (a per-lane 64-bit
            // pointer here was one of the register pairs the C2 modes kept in scratch)
            const uint32_t tb1_off = (uint32_t)(aoff_of(AX_B, aoff) + o0 * a_c0[AX_B]) * 8u;
            const uint32_t tb1_step = (uint32_t)a_c1[AX_B] * 8u;
            gptr<char> tb1_base = reinterpret_cast<gptr<char>>(atab[AX_B]);
            auto tb1_at = [&](uint32_t byte_step) __attribute__((always_inline)) -> i2v {
                return *reinterpret_cast<gptr<i2v>>(tb1_base + (tb1_off + byte_step));
            };
            i2v e_nx = {0, 0};
            float g_nx = 0.f;
            // cost so far + the level-1 term; when that term is the first of the sum, -0 + x == x bit for bit (no select per step)
            const float go0_l1 = cl1_first ? -0.0f : go0;
            auto level1_cost = [&](int o1) -> float {
                if (!cl1_present) return go0;
                return go0_l1 + cterm(CL1, o0, o1);
            };
            // ... of steps o1 and o1 + 1 as a pair: the two LDS reads are issued at the top of a trip and added where the sweep starts
            // (an absent term reads the -0 row of s_r2: no branch, no merge)
            const float *l1_row = cl1_present ? s_ot + cl_off[1] + o0 * cl_c0[1] : reinterpret_cast<const float *>(s_r2 + npairs);
            const int l1_step = cl1_present ? cl_c1[1] : 0;
            auto level1_terms2 = [&](int o1) -> f2 { return (f2){l1_row[o1 * l1_step], l1_row[(o1 + 1) * l1_step]}; };
            i2v e_nx2 = {0, 0};                   // ... and the entry after it: a two-step trip consumes two
            if constexpr (HIER) {
                e_nx = tb1_at(0u);
                if (m_o1 > 1) e_nx2 = tb1_at(tb1_step);
                g_nx = level1_cost(0);
            }
            // base offset of the outer axes' cells: modes 1-3 need it on the rare paths only
            auto outer_base = [&]() {
                int b = 0;
#pragma unroll
                for (int a = 0; a < D - 1; ++a) b += js[a] * cell[a];
                return b;
            };
            int o1 = 0;
            // (the window modes only: at the C2 modes' 96-register budget the eight kept values cost nine spills to
            // scratch - 40 MB of write-back per C2 stage - and buy nothing measurable there)
            constexpr bool KEEP_ROWS = PRE;
            int cell_rows = -1;                                          // the level-1 cell the kept rows were selected for (none yet)
            float R0[4], RD[4];
            // ---- hierarchical modes, TWO (o0, o1) steps per trip --------------------------------------------
            // When every cell change of the wave is a wave-uniform first crossing (UX == 0: the C2 and attitude
            // shapes) and both steps stay inside the prepared window, steps o1 and o1+1 share one pass over the
            // controls: one read of (t, r2) per control pair, one set of loop bookkeeping, two running minima.  Per
            // element the arithmetic is unchanged, and step o1 is compared with `best` before step o1+1: same bits,
            // same first-minimum.  Anything else falls through to the one-step loop below.
            if constexpr (HIER) {
                if (UX == 0u && !slow_a) {
                    // the two steps' entries are carried from trip to trip and reloaded IN PLACE once the trip has used them (index clamped
                    // to the last step: no test, nothing to merge): no "next" copies to rotate at the end of a trip
                    i2v eA = e_nx, eB = e_nx2;
                    const int o1_last = m_o1 - 1;
                    while (HJB_K3_PROBE != 1 && o1 + 1 < m_o1) {
                        // (E0, dE) of the first / second last-axis cell, as {step A, step B} pairs.  The two prepared rows a step
                        // lerps between, R0 = F[r] and RD = F[r+1] - F[r], are kept from trip to trip (r moves 0 -> 1 at most
                        // once over an o1 sweep when the axis' next value grows with the control): while every lane's cell of both
                        // steps is the one the rows were selected for - two compares; that also says "inside the window" - the
                        // lerps are 4 packed fmas instead of 16 selects + 4 packed subtractions + 4 packed fmas.  Same f1 - f0, same
                        // fma: same bits.
                        const f2 c2 = level1_terms2(o1);
                        bool kept = KEEP_ROWS && !__any(eA.x != cell_rows || eB.x != cell_rows);
                        const int rA = eA.x - c1min, rB = eB.x - c1min;
                        if (!kept) {
                            if (__any(((unsigned int)rA | (unsigned int)rB) > 1u)) break;   // outside the window: eA still belongs to o1
                            // The kept rows are rewritten IN PLACE, in a block of its own, when both steps sit on the same row;
                            // the trip that straddles a move (step A on one row pair, step B on the next) selects per step and leaves
                            // them alone.  (Written as "either path assigns R0 / RD" the compiler put nine register copies on the common
                            // path to merge the two definitions: 5 % of a 6-D state's vector instructions.)
                            if (KEEP_ROWS && !__any(rA != rB)) {
                                const bool up = rA != 0;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float f0 = up ? F[1][q] : F[0][q];
                                    const float f1 = up ? F[2][q] : F[1][q];
                                    R0[q] = f0;
                                    RD[q] = f1 - f0;
                                }
                                cell_rows = eA.x;
                                kept = true;
                            }
                        }
                        const f2 t2 = {__int_as_float(eA.y), __int_as_float(eB.y)};
                        f2 X2[4];
                        if (kept) {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                X2[q] = __builtin_elementwise_fma(t2, (f2){RD[q], RD[q]}, (f2){R0[q], R0[q]});
                        } else {
                            const bool upA = rA != 0, upB = rB != 0;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f2 f0 = {upA ? F[1][q] : F[0][q], upB ? F[1][q] : F[0][q]};
                                const f2 f1 = {upA ? F[2][q] : F[1][q], upB ? F[2][q] : F[1][q]};
                                const f2 d = f1 - f0;
                                X2[q] = __builtin_elementwise_fma(t2, d, f0);
                            }
                        }
                        {
                            const int oa = o1 + 2 < o1_last ? o1 + 2 : o1_last, ob = o1 + 3 < o1_last ? o1 + 3 : o1_last;
                            eA = tb1_at((uint32_t)oa * tb1_step);
                            eB = tb1_at((uint32_t)ob * tb1_step);
                        }
                        // The two halves of every packed instruction are the two STEPS (A, B) of one control: the per-step
                        // quantities (cost so far, E0, dE) are register pairs as they come out of the lerps above, the
                        // per-control ones (t_j, r_j) are broadcast by the instruction's operand selects, so nothing has to be
                        // rearranged; a control pair is 6 packed instructions + 2 v_min3.  UX == 0 leaves at most ONE cell
                        // change in the sweep, the whole wave's, at control jc: controls before it use the first cell, the
                        // others the second - three straight loops, no per-pair tests.
                        const f2 Ea = X2[0], Da = X2[1] - X2[0], Eb = X2[2], Db = X2[3] - X2[2];
                        const f2 g2 = (f2){go0_l1, go0_l1} + c2;                 // cost so far of {step A, step B}
                        float mA = INFINITY, mB = INFINITY;
                        const int pa = jc >> 1;                                  // pairs wholly in the first cell
                        const int nfull = m_in >> 1;
#ifndef HJB_K3_UNROLLED_PAIRS
#define HJB_K3_UNROLLED_PAIRS 1
#endif
                        if (HJB_K3_UNROLLED_PAIRS && (PRE || (M1 && (m_in & 1))) && npairs == kPairsUnrolled) {
                            // The usual sweeps - the window modes' 11 or 12 inner controls (6 pair rows; C3, the 6-D grids) and modes 1 / 4's 21
                            // (11 pair rows; C2) - are written out STRAIGHT-LINE for the trip's shape - the full pairs (+ one control alone), how many pairs
                            // lie wholly in the first cell, whether the wave's one cell change splits a pair or falls on the control that
                            // stands alone: one scalar jump per trip picks the sequence.  With the pair number a compile-time constant the
                            // LDS rows of (t, r) are read at immediate offsets from two fixed address registers, and with no branch inside
                            // the sequence nothing is copied at a join: 8 vector instructions per pair and step pair (6 packed + 2 v_min3)
                            // instead of ~10.5, and the control that stands alone costs 3 packed + 2 v_min with its cell picked at compile time.
                            auto pairs_fixed = [&](auto NFc, auto PAc, auto STc, auto LSc) __attribute__((always_inline)) {
                                constexpr int NF = decltype(NFc)::value, PA = decltype(PAc)::value;
                                constexpr bool ST = decltype(STc)::value, LS = decltype(LSc)::value;
                                // (t, r) of a pair row are read one row ahead, into a ring of registers the unrolled sequence names statically
                                f2 tr[3], rr[3];
                                tr[0] = lds_f2(my_t);
                                rr[0] = lds_f2(s_r2);
                                tr[1] = lds_f2(my_t + 256);
                                rr[1] = lds_f2(s_r2 + 1);
                                // (a marker that differs per sequence: the four reads above may be hoisted in front of the jump, the arithmetic
                                // below may not - hoisted, its {t, t} operand pairs are built with moves in the jump's block instead of being
                                // folded into the packed instructions' operand selects: 6 moves per trip)
                                asm volatile("; trip shape %2" : "+v"(tr[0]), "+v"(tr[1]) : "n"(NF * 64 + PA * 4 + (ST ? 2 : 0) + (LS ? 1 : 0)));
#pragma unroll
                                for (int q = 0; q < NF; ++q) {
                                    constexpr int kLast = kPairsUnrolled - 1;
                                    const int qn = q + 2 < kLast ? q + 2 : kLast;              // (compile-time after unrolling)
#ifndef HJB_K3_PREFETCH2
#define HJB_K3_PREFETCH2 0
#endif
                                    if (HJB_K3_PREFETCH2) {              // (A/B: two rows ahead - slower, r04_k3_experiments.log)
                                        tr[(q + 2) % 3] = lds_f2(my_t + qn * 256);
                                        rr[(q + 2) % 3] = lds_f2(s_r2 + (q + 2 < kPairsUnrolled ? q + 2 : kPairsUnrolled));
                                    } else if (q > 0 && q + 1 <= kLast) {
                                        tr[(q + 1) % 3] = lds_f2(my_t + (q + 1) * 256);
                                        rr[(q + 1) % 3] = lds_f2(s_r2 + q + 1);
                                    }
                                    const f2 tc = tr[q % 3], rc = rr[q % 3];
                                    const f2 Ex = (q < PA || (ST && q == PA)) ? Ea : Eb, Dx = (q < PA || (ST && q == PA)) ? Da : Db;
                                    const f2 Ey = q < PA ? Ea : Eb, Dy = q < PA ? Da : Db;
                                    const f2 totx = (g2 + (f2){rc.x, rc.x}) + __builtin_elementwise_fma((f2){tc.x, tc.x}, Dx, Ex);
                                    const f2 toty = (g2 + (f2){rc.y, rc.y}) + __builtin_elementwise_fma((f2){tc.y, tc.y}, Dy, Ey);
                                    mA = __builtin_fminf(__builtin_fminf(mA, totx.x), toty.x);
                                    mB = __builtin_fminf(__builtin_fminf(mB, totx.y), toty.y);
                                }
                                if constexpr (NF < kPairsUnrolled) {                             // 11 controls: the last one stands alone (row NF, first half)
                                    constexpr bool second = PA < NF || LS;                       // ... in the second cell when the change came before or at it
                                    const f2 tc = tr[NF % 3], rc = rr[NF % 3];
                                    const f2 El = second ? Eb : Ea, Dl = second ? Db : Da;
                                    const f2 totx = (g2 + (f2){rc.x, rc.x}) + __builtin_elementwise_fma((f2){tc.x, tc.x}, Dl, El);
                                    mA = __builtin_fminf(mA, totx.x);
                                    mB = __builtin_fminf(mB, totx.y);
                                }
                            };
                            // (wave-uniform by construction - jc comes from ballots; readfirstlane says so to the compiler - and a scalar
                            // of THIS trip: one jump, not a chain of hoisted lane masks)
                            int sel = __builtin_amdgcn_readfirstlane(nfull * 64 + (pa < nfull ? pa : nfull) * 4 + (((jc & 1) && jc < m_in) ? 2 : 0)
                                                                     + (((m_in & 1) && jc == m_in - 1) ? 1 : 0));
                            asm volatile("" : "+s"(sel));
#define HJB_PF(NF, PA, ST, LS) case (NF) * 64 + (PA) * 4 + (ST) * 2 + (LS): pairs_fixed(std::integral_constant<int, NF>{}, std::integral_constant<int, PA>{}, \
                                       std::integral_constant<bool, (ST) != 0>{}, std::integral_constant<bool, (LS) != 0>{}); break;
                            if (HJB_K3_PROBE == 3) {
                                sel = -1;
                                mA = __builtin_fminf(Ea.x + Da.x, Eb.x + Db.x) + g2.x;
                                mB = __builtin_fminf(Ea.y + Da.y, Eb.y + Db.y) + g2.y;
                            }
                            if constexpr (M1) {                                // 21 controls: ten full pairs + one control alone
                                switch (sel) {
                                    HJB_PF(10, 0, 0, 0) HJB_PF(10, 0, 1, 0) HJB_PF(10, 1, 0, 0) HJB_PF(10, 1, 1, 0) HJB_PF(10, 2, 0, 0) HJB_PF(10, 2, 1, 0)
                                    HJB_PF(10, 3, 0, 0) HJB_PF(10, 3, 1, 0) HJB_PF(10, 4, 0, 0) HJB_PF(10, 4, 1, 0) HJB_PF(10, 5, 0, 0) HJB_PF(10, 5, 1, 0)
                                    HJB_PF(10, 6, 0, 0) HJB_PF(10, 6, 1, 0) HJB_PF(10, 7, 0, 0) HJB_PF(10, 7, 1, 0) HJB_PF(10, 8, 0, 0) HJB_PF(10, 8, 1, 0)
                                    HJB_PF(10, 9, 0, 0) HJB_PF(10, 9, 1, 0) HJB_PF(10, 10, 0, 0) HJB_PF(10, 10, 0, 1)
                                    default:
                                        if (HJB_K3_PROBE != 3) __builtin_unreachable();
                                        break;
                                }
                            } else {
                                switch (sel) {
                                    HJB_PF(5, 0, 0, 0) HJB_PF(5, 0, 1, 0) HJB_PF(5, 1, 0, 0) HJB_PF(5, 1, 1, 0) HJB_PF(5, 2, 0, 0) HJB_PF(5, 2, 1, 0)
                                    HJB_PF(5, 3, 0, 0) HJB_PF(5, 3, 1, 0) HJB_PF(5, 4, 0, 0) HJB_PF(5, 4, 1, 0) HJB_PF(5, 5, 0, 0) HJB_PF(5, 5, 0, 1)
                                    HJB_PF(6, 0, 0, 0) HJB_PF(6, 0, 1, 0) HJB_PF(6, 1, 0, 0) HJB_PF(6, 1, 1, 0) HJB_PF(6, 2, 0, 0) HJB_PF(6, 2, 1, 0)
                                    HJB_PF(6, 3, 0, 0) HJB_PF(6, 3, 1, 0) HJB_PF(6, 4, 0, 0) HJB_PF(6, 4, 1, 0) HJB_PF(6, 5, 0, 0) HJB_PF(6, 5, 1, 0)
                                    HJB_PF(6, 6, 0, 0)
                                    default:               // npairs == 6 means 5 or 6 full pairs, and the other three fields follow from jc
                                        if (HJB_K3_PROBE != 3) __builtin_unreachable();
                                        break;
                                }
                            }
#undef HJB_PF
                        } else {
                            f2 t = my_t[0];
                            f2 r2 = s_r2[0];
                            auto control_pair = [&](int p, const f2 &Ex, const f2 &Dx, const f2 &Ey, const f2 &Dy)
                                                    __attribute__((always_inline)) {
                                const f2 totx = (g2 + (f2){r2.x, r2.x}) + __builtin_elementwise_fma((f2){t.x, t.x}, Dx, Ex);
                                const f2 toty = (g2 + (f2){r2.y, r2.y}) + __builtin_elementwise_fma((f2){t.y, t.y}, Dy, Ey);
                                t = lds_f2(my_t + t_ahead(p) * 256);                 // next pair
                                r2 = lds_f2(s_r2 + p + 1);
                                // left-leaning chains: the instruction selector folds min(min(m, x), y) into ONE v_min3_f32 per pair; written
                                // min(m, min(x, y)) it pairs the inner minima of two pairs instead (3 instructions per two values)
                                mA = __builtin_fminf(__builtin_fminf(mA, totx.x), toty.x);
                                mB = __builtin_fminf(__builtin_fminf(mB, totx.y), toty.y);
                            };
                            int p = 0;
#pragma unroll 2
                            for (; p < pa; ++p) control_pair(p, Ea, Da, Ea, Da);
                            if ((jc & 1) && jc < m_in) {                         // the change falls on a pair's second control
                                control_pair(p, Ea, Da, Eb, Db);
                                ++p;
                            }
#pragma unroll 2
                            for (; p < nfull; ++p) control_pair(p, Eb, Db, Eb, Db);
                            if (m_in & 1) {                                      // the last control of an odd sweep, alone
                                const bool second = jc < m_in;                   // (wave-uniform) it lies in the second cell
                                const f2 El = second ? Eb : Ea, Dl = second ? Db : Da;
                                const f2 totx = (g2 + (f2){r2.x, r2.x}) + __builtin_elementwise_fma((f2){t.x, t.x}, Dl, El);
                                mA = __builtin_fminf(mA, totx.x);
                                mB = __builtin_fminf(mB, totx.y);
                            }
                        }
                        const int uo = uo0 + o1;
                        if (uo == 0 || mA < best) { best = mA; best_uo = uo; }
                        if (mB < best) { best = mB; best_uo = uo + 1; }
                        o1 += 2;
                    }
                    e_nx = eA;                                                   // the one-step loop below takes over at o1
                    if (o1 < m_o1) g_nx = level1_cost(o1);
                }
            }
            if (HJB_K3_PROBE == 1) {
                for (int rb = 0; rb < (HIER ? 3 : 1); ++rb) for (int q = 0; q < 4; ++q) best = __builtin_fminf(best, F[rb][q]);
            }
            for (; HJB_K3_PROBE != 1 && !(HJB_K3_PROBE == 4 && o1 == m_o1 - 1 && o1 > 0) && o1 < m_o1; ++o1) {
                const int uo = uo0 + o1;
                // ---- level 1: (E0, dE) of the two last-axis cells this state visits ----------
                float e0a, dea, e0b, deb;
                if constexpr (HIER) {
                    cell[AX_B] = e_nx.x;
                    tw[AX_B] = __int_as_float(e_nx.y);
                    go = g_nx;
                    if (o1 + 1 < m_o1) {
                        e_nx = tb1_at((uint32_t)(o1 + 1) * tb1_step);
                        g_nx = level1_cost(o1 + 1);
                    }
                    const int r = cell[AX_B] - c1min;
                    if (KEEP_ROWS && !__any(cell[AX_B] != cell_rows)) {  // the rows the trips kept (hence inside the window): 4 fmas
                        const float t1 = tw[AX_B];
                        float X[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) X[q] = __builtin_fmaf(t1, RD[q], R0[q]);
                        e0a = X[0]; dea = X[1] - X[0];
                        e0b = X[2]; deb = X[3] - X[2];
                    } else if ((r == 0 || r == 1) && !slow_a) {          // inside the prepared 2-cell window
                        const bool up = r != 0;
                        const float t1 = tw[AX_B];
                        float X[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float f0 = up ? F[1][q] : F[0][q];
                            const float f1 = up ? F[2][q] : F[1][q];
                            X[q] = __builtin_fmaf(t1, f1 - f0, f0);
                        }
                        e0a = X[0]; dea = X[1] - X[0];
                        e0b = X[2]; deb = X[3] - X[2];
                    } else {                                             // rare: axis 1 spans > 2 cells in this sweep
                        const int ob = outer_base();
                        cell_pair(ob, lc0, tw, e0a, dea);
                        cell_pair(ob, lc1, tw, e0b, deb);
                    }
                } else {
                    f2 v[1 << D];                                    // {first cell, second cell} contracted together
#pragma unroll
                    for (int c = 0; c < (1 << D); ++c) v[c] = (f2){G[c], H[c]};
#pragma unroll
                    for (int a = 0; a < D - 1; ++a) {
                        const f2 w = {twc[a], twc[a]};
#pragma unroll
                        for (int jj = 0; jj < (1 << (D - 1 - a)); ++jj)
                            v[jj] = __builtin_elementwise_fma(w, v[2 * jj + 1] - v[2 * jj], v[2 * jj]);
                    }
                    const f2 d = v[1] - v[0];
                    e0a = v[0].x; e0b = v[0].y;
                    dea = d.x;    deb = d.y;
                }
                const bool has_next = !HIER && o1 + 1 < m_o1;
                if constexpr (!HIER) {
                    if (has_next) {
                        prepare(o1 + 1, base_n, twn, go_n);
                        load_corners<D>(Jn, base_n + js[D - 1] * lc0, js, G);
                        load_corners<D>(Jn, base_n + js[D - 1] * lc1, js, H);
                    }
                }
                float e0 = e0a, de = dea;                            // (E0, dE) of the cell the query is in
                const f2 go2 = {go, go};
                float ibest = INFINITY;
                // new (E0, dE) when this lane's query enters another cell at control j
                auto crossed = [&](int j, float &ne0, float &nde) {
                    const unsigned int first = cm & (0u - cm);
                    if ((first >> j) & 1u) {             // first crossing: corners were prefetched
                        ne0 = e0b;
                        nde = deb;
                    } else {                             // later crossings: general path
                        const float q = ql + (b_pure ? s_b[j] : b_data[boff + j * b_stride]);
                        int lc = find_cell<float>(s_k, nl, q, l_uniform, l_x0, l_invh) - plane0;
                        if (lc < 0 || lc + 1 >= nplanes) {
                            *P->status = 1;
                            lc = lc < 0 ? 0 : nplanes - 2;
                        }
                        if constexpr (HIER) cell_pair(outer_base(), lc, tw, ne0, nde);
                        else cell_pair(base, lc, twc, ne0, nde);
                    }
                };
                // ---- inner loop over control PAIRS: packed, no loads, no cell tests on the vector pipe.
                // Per pair: 3 packed math (E0, dE broadcast to both halves) and one v_min3 (minima only: the label is found
                // after the sweep).
                // The pair loop is cut at the (wave-uniform, scalar) pairs where some lane changes cell, so
                // that inside a segment (E0, dE) are loop-invariant registers.
                f2 t = my_t[0];
                f2 r2 = s_r2[0];
                int p = 0;
                if (HJB_K3_UNROLLED_PAIRS && PRE && UX == 0u && npairs == kPairsUnrolled) {
                    // The window modes' usual sweep, one step (the last of an odd step count): as in the two-step trip above the six
                    // control pairs run STRAIGHT-LINE for the wave's one cell change at control jc - pairs below jc >> 1 in the first
                    // cell, the pair jc splits (jc odd) half and half, the rest in the second cell; a padding control (odd counts)
                    // carries an infinite r and never wins.  One scalar jump per step, LDS rows at immediate offsets.
                    auto step_fixed = [&](auto PAc, auto STc) __attribute__((always_inline)) {
                        constexpr int PA = decltype(PAc)::value;
                        constexpr bool ST = decltype(STc)::value;
#pragma unroll
                        for (int q = 0; q < kPairsUnrolled; ++q) {
                            const f2 dv = q < PA ? (f2){dea, dea} : ((ST && q == PA) ? (f2){dea, deb} : (f2){deb, deb});
                            const f2 ev = q < PA ? (f2){e0a, e0a} : ((ST && q == PA) ? (f2){e0a, e0b} : (f2){e0b, e0b});
                            const f2 tot = (go2 + r2) + __builtin_elementwise_fma(t, dv, ev);
                            t = lds_f2(my_t + t_ahead(q) * 256);
                            r2 = lds_f2(s_r2 + q + 1);
                            ibest = __builtin_fminf(__builtin_fminf(ibest, tot.x), tot.y);
                        }
                    };
                    int sel1 = __builtin_amdgcn_readfirstlane(jc < 2 * kPairsUnrolled ? jc : 2 * kPairsUnrolled);      // (jc >> 1) * 2 + (jc & 1)
                    asm volatile("" : "+s"(sel1));
#define HJB_SF(J) case (J): step_fixed(std::integral_constant<int, (J) / 2>{}, std::integral_constant<bool, ((J) & 1) != 0>{}); break;
                    switch (sel1) {
                        HJB_SF(1) HJB_SF(2) HJB_SF(3) HJB_SF(4) HJB_SF(5) HJB_SF(6) HJB_SF(7) HJB_SF(8) HJB_SF(9) HJB_SF(10) HJB_SF(11) HJB_SF(12)
                        default: break;            // unreachable: 1 <= jc (bit 0 of the crossing mask is never set)
                    }
#undef HJB_SF
                    p = npairs;
                }
                while (p < npairs) {
                    const unsigned int rest = PU >> p;
                    const int pstop = rest ? p + __builtin_ctz(rest) : npairs;   // next pair with a crossing
                    const f2 e0v = {e0, e0}, dev = {de, de};
#pragma unroll 2
                    for (; p < pstop; ++p) {
                        const f2 tot = (go2 + r2) + __builtin_elementwise_fma(t, dev, e0v);
                        t = lds_f2(my_t + t_ahead(p) * 256);         // next pair's rows
                        r2 = lds_f2(s_r2 + p + 1);
                        ibest = __builtin_fminf(__builtin_fminf(ibest, tot.x), tot.y);            // v_min3_f32
                    }
                    if (p < npairs) {                                // a pair in which some lane changes cell
                        const int jb = 2 * p;
                        float e0y, dey;
                        if (((UX >> jb) & 3u) == 0u) {               // wave-uniform first crossings only: scalar
                            if ((UA >> jb) & 1u) { e0 = e0b; de = deb; }
                            e0y = e0;
                            dey = de;
                            if ((UA >> (jb + 1)) & 1u) { e0y = e0b; dey = deb; }
                        } else {
                            if ((cm >> jb) & 1u) crossed(jb, e0, de);
                            e0y = e0;
                            dey = de;
                            if ((cm >> (jb + 1)) & 1u) crossed(jb + 1, e0y, dey);
                        }
                        const f2 tot = (go2 + r2) + __builtin_elementwise_fma(t, (f2){de, dey}, (f2){e0, e0y});
                        e0 = e0y;
                        de = dey;
                        t = my_t[t_ahead(p) * 256];
                        r2 = s_r2[p + 1];
                        ibest = __builtin_fminf(__builtin_fminf(ibest, tot.x), tot.y);
                        ++p;
                    }
                }
                if (uo == 0 || ibest < best) {                       // which inner control: resolved after the sweep
                    best = ibest;
                    best_uo = uo;
                }
                if (has_next) {
                    base = base_n;
#pragma unroll
                    for (int a = 0; a < D - 1; ++a) twc[a] = twn[a];
                    go = go_n;
                }
            }  // o1
        }      // o0
        // ---- which inner control?  The sweep keeps only the minimum of every (o0, o1) step (one v_min3 per control pair and
        // step, no compare / select) and the first step that attains the overall minimum.  That step's controls are now
        // re-evaluated in order exactly as the sweep did (same cells, same weights, same lerp order, same sums): the first
        // one that reproduces `best` wins (first-minimum rule).  Once per state instead of once per pair and step.
        int best_j = 0;
        {
            const int o0 = (int)udiv_gm((uint32_t)best_uo, N->div_m_o1, N->div_s_o1), o1 = best_uo - o0 * m_o1;
            int ob = 0;
#pragma unroll
            for (int a = 0; a < D - 1; ++a) {
                if (a_lvl(a) >= 0) {
                    if (INL0 && a == 0) {
                        axis0_entry(qs0, o0, cell[a], tw[a]);
                    } else {
                        const i2v e = at_ld(a, aoff_of(a, aoff) + o0 * a_c0[a] + (a_lvl(a) == 1 ? o1 * a_c1[a] : 0));
                        cell[a] = e.x;
                        tw[a] = __int_as_float(e.y);
                    }
                }
                ob += js[a] * cell[a];
            }
            float g = gpre;
            if (cl0_present) {
                const float x = cterm(CL0, o0, 0);
                g = cl0_first ? x : g + x;
            }
            if (cl1_present) {
                const float x = cterm(CL1, o0, o1);
                g = cl1_first ? x : g + x;
            }
            // (E0, dE) of last-axis cell lc at this step's outer cells: from the state's LDS window when it covers them
            auto step_cell = [&](int lc, float &xe0, float &xde) {
                if constexpr (PRE) {
                    const int ra = cell[AX_A] - cAmin, rb = cell[AX_B] - cBmin;
                    if ((ra == 0 || ra == 1) && (rb == 0 || rb == 1) && (lc == lc0 || lc == lc1) && !far_cells) {
                        const int q0 = W3P ? lc - pl0 : (lc == lc0 ? 0 : 2);
                        float X[2];
#pragma unroll
                        for (int dq = 0; dq < 2; ++dq) {
                            float Fr[2];
#pragma unroll
                            for (int db = 0; db < 2; ++db) {
                                const float w0 = my_w[((ra * 3 + rb + db) * kWq + q0 + dq) * 256];
                                const float w1 = my_w[(((ra + 1) * 3 + rb + db) * kWq + q0 + dq) * 256];
                                Fr[db] = __builtin_fmaf(tw[AX_A], w1 - w0, w0);
                            }
                            X[dq] = __builtin_fmaf(tw[AX_B], Fr[1] - Fr[0], Fr[0]);
                        }
                        xe0 = X[0];
                        xde = X[1] - X[0];
                        return;
                    }
                }
                cell_pair(ob, lc, tw, xe0, xde);
            };
            float xe0, xde;
            step_cell(lc0, xe0, xde);
            int ncross = 0;
#pragma unroll 1
            for (int j = 0; j < m_in; ++j) {
                if (j > 0 && ((cm >> j) & 1u)) {                     // this lane's query enters another cell at control j
                    int lc = lc1;                                    // first crossing: the second prefetched cell
                    if (ncross++ > 0) {
                        const float q = ql + (b_pure ? s_b[j] : b_data[boff + j * b_stride]);
                        lc = find_cell<float>(s_k, nl, q, l_uniform, l_x0, l_invh) - plane0;
                        if (lc < 0 || lc + 1 >= nplanes) {
                            *P->status = 1;
                            lc = lc < 0 ? 0 : nplanes - 2;
                        }
                    }
                    step_cell(lc, xe0, xde);
                }
                const f2 tp = my_t[(j >> 1) * 256], rp = s_r2[j >> 1];
                const float tot = (g + ((j & 1) ? rp.y : rp.x)) + __builtin_fmaf((j & 1) ? tp.y : tp.x, xde, xe0);
                if (tot == best) {
                    best_j = j;
                    break;
                }
            }
        }
        if (valid) {
            int label;
            if (C == 1) {
                label = best_j;
            } else if (C == 2) {
                label = best_uo + P->m[0] * best_j;
            } else {
                const int j0 = (int)udiv_gm((uint32_t)best_uo, N->div_m_o1, N->div_s_o1), j1 = best_uo - j0 * P->m[1];     // m_o1 == m[1] when C == 3
                label = j0 + P->m[0] * (j1 + P->m[1] * best_j);
            }
            sidx_t in_plane, pl;
            if constexpr (QMODEL) {
                in_plane = ls % inner_sz; pl = ls / inner_sz;
            } else {
                pl = (sidx_t)udiv_gm((uint32_t)ls, N->div_m_inner, N->div_s_inner);
                in_plane = ls - pl * inner_sz;
            }
            Jout[in_plane + (sidx_t)inner_sz * (pl + P->halo_lo)] = (TJ)best;
            if (idx_out) st_idx(idx_out, ls, label + P->index_base, P->idx_bytes);
        }
    }
}

}  // namespace hjb
