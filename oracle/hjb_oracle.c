/* hjb_oracle.c - CPU restatement (plain C + OpenMP) of the reference's Bellman
 * backup.  TEST INFRASTRUCTURE ONLY: it is the checker for the HIP path and the
 * timed "cpu_baseline" of bench.py; the product (libhjbdp) never links or calls it.
 *
 * Follows, per stage,
 *     [F.Values, idx] = min( J_stage + F(x_next_1,...,x_next_D), [], ctrl_dim )
 *   test/Dynamic_Solver.m:207-210, test/test_coder.m:28-36,109-117,
 *   position-control/Solver_position.m:135-137,
 *   attitude-control/Solver_attitude.m:239-241,400-409,
 *   pos-att/Solver_pos_att.m:272 (+ early-stop monitor :268-285)
 * with F = griddedInterpolant(...,'linear') (N-linear, linear extrapolation) and
 * `min` returning the first minimal index (cascade order for C > 1).
 *
 * Parity pin: oracle/hjb_oracle.py (same algorithm, numpy) reproduces
 * test/obj_1.mat to 7.4e-14; tests/test_oracle_golden.py checks this C twin
 * against the same fixture and against the numpy version.
 *
 * This twin fixes the floating-point evaluation order ("canonical arithmetic")
 * so that the HIP kernels can be compared BIT-FOR-BIT:
 *   q_a   = ((t0 + t1) + t2) ...                 plain adds, left to right
 *   cell  = clamp(upper_bound(knots,q) - 1, 0, n-2)          exact search
 *   t     = (q - k[cell]) * rdx[cell],  rdx = 1/(k[i+1]-k[i]) rounded to dtype
 *   lerp  = fma(t, v1 - v0, v0), axis 0 first ... axis D-1 last
 *   total = g + interp,  g = ((c0 + c1) + c2) ...
 *   argmin: strict '<' while visiting controls with control dim 0 SLOWEST
 *           (= cascade min over dims D+C-1, ..., D of Solver_attitude.m:400-409)
 * Compile with -ffp-contract=off so only the explicit fma() contracts.
 *
 * hjb_problem.table_dtype == HJB_TAB_F64 (float32 problems): the reference's pos-att typing, Solver_pos_att.m:299-327 -
 * next-state terms are float64 arrays; q, the cell search (on the float64 knots as given) and the weight
 * (q - k[c]) * (1 / (k[c+1] - k[c])) are float64, the weight is rounded to float32 once; blend and cost stay float32.
 * Argmin labels are always written as int32 here, whatever hjb_problem.idx_dtype says (the wrapper widens the
 * library's labels before comparing).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE            /* sched_getaffinity, CPU_COUNT */
#endif
#include <immintrin.h>
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/hjbdp.h"

typedef struct {
    const void *data;
    int64_t stride[HJB_MAX_G];
    int has_ctrl;
} term_t;

static void build_term(const hjb_problem *p, const hjb_term *t, term_t *out) {
    int G = p->D + p->C;
    int64_t s = 1;
    out->data = t->data;
    out->has_ctrl = 0;
    for (int d = 0; d < HJB_MAX_G; ++d) out->stride[d] = 0;
    for (int d = 0; d < G; ++d) {
        if (t->mask & (1u << d)) {
            out->stride[d] = s;
            s *= (d < p->D) ? p->n[d] : p->m[d - p->D];
            if (d >= p->D) out->has_ctrl = 1;
        }
    }
}

static int validate(const hjb_problem *p) {
    if (!p || p->D < 1 || p->D > HJB_MAX_D || p->C < 1 || p->C > HJB_MAX_C) return HJB_E_INVALID;
    if (p->dtype != HJB_F32 && p->dtype != HJB_F64 && p->dtype != HJB_F16S) return HJB_E_UNSUPPORTED;
    for (int a = 0; a < p->D; ++a) {
        if (p->n[a] < 2 || !p->knots[a]) return HJB_E_INVALID;
        const int model_axis = p->model == HJB_MODEL_QUAT_EULER321 && a < 3;
        if (model_axis ? p->n_next_terms[a] != 0 : (p->n_next_terms[a] < 1 || p->n_next_terms[a] > HJB_MAX_TERMS))
            return HJB_E_INVALID;
    }
    if (p->model != HJB_MODEL_NONE && p->model != HJB_MODEL_QUAT_EULER321) return HJB_E_INVALID;
    if (p->model == HJB_MODEL_QUAT_EULER321) {
        if (p->D != 6 || p->C != 3 || p->dtype == HJB_F64) return HJB_E_UNSUPPORTED;
        for (int i = 0; i < 4; ++i)
            if (!p->model_tables[i]) return HJB_E_INVALID;
    }
    for (int c = 0; c < p->C; ++c)
        if (p->m[c] < 1) return HJB_E_INVALID;
    if (p->n_cost_terms < 1 || p->n_cost_terms > HJB_MAX_TERMS) return HJB_E_INVALID;
    return HJB_OK;
}

/* ---- HJB_MODEL_QUAT_EULER321: next (yaw, pitch, roll) of one state ------------------------------
 * attitude-control/Solver_attitude.m:449-489 in single precision, operation by operation: Euler step of
 * the quaternion kinematics (:449-467), renormalisation (:477-483), back to Euler angles (:485-489).
 * MATLAB's atan2/asin are closed source; the library (and therefore this restatement) uses fixed
 * polynomial forms built from +,-,*,/ and sqrt only, so that CPU and GPU agree bit for bit.  They are
 * within a few ulp of libm (tests/test_oracle_golden.py checks that). */
static float canon_atan2f(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const int swap = ay > ax;
    const float num = swap ? ax : ay, den = swap ? ay : ax;
    float r = den == 0.0f ? 0.0f : num / den;               /* in [0, 1] */
    float off = 0.0f;
    if (r > 0.4142135623730950f) {                           /* tan(pi/8) */
        r = (r - 1.0f) / (r + 1.0f);
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

static float canon_asinf(float x) {
    const float a = fabsf(x);
    const int big = a > 0.5f;
    float z, r;
    if (big) {
        z = 0.5f * (1.0f - a);
        r = sqrtf(z);
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

void orc_canon_eval(int kind, int64_t n, const float *a, const float *b, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = kind == 0 ? canon_atan2f(a[i], b[i]) : canon_asinf(a[i]);
}

/* gi: grid indices of the state; W = knots of axes 3..5 (single) */
static void model_quat_next(const hjb_problem *p, const int *gi, float w1, float w2, float w3, float *out3) {
    const int64_t ti = gi[0] + (int64_t)p->n[0] * (gi[1] + (int64_t)p->n[1] * gi[2]);
    const float q1 = ((const float *)p->model_tables[0])[ti], q2 = ((const float *)p->model_tables[1])[ti];
    const float q3 = ((const float *)p->model_tables[2])[ti], q7 = ((const float *)p->model_tables[3])[ti];
    const float h = (float)p->model_h, half = 0.5f;
    float x4 = q1 + h * (half * ((w3 * q2 - w2 * q3) + w1 * q7));       /* :449-452 */
    float x5 = q2 + h * (half * ((-w3 * q1 + w1 * q3) + w2 * q7));      /* :454-457 */
    float x6 = q3 + h * (half * ((w2 * q1 - w1 * q2) + w3 * q7));       /* :459-462 */
    float x7 = q7 + h * (half * ((-w1 * q1 - w2 * q2) - w3 * q3));      /* :465-467 */
    const float nrm = sqrtf(((x4 * x4 + x5 * x5) + x6 * x6) + x7 * x7); /* :477 */
    x4 = x4 / nrm; x5 = x5 / nrm; x6 = x6 / nrm; x7 = x7 / nrm;         /* :480-483 */
    out3[0] = canon_atan2f(2.0f * (x6 * x5 + x7 * x4), ((x7 * x7 + x6 * x6) - x5 * x5) - x4 * x4);   /* :485-486 */
    out3[1] = canon_asinf(-2.0f * (x6 * x4 - x7 * x5));                                              /* :487 */
    out3[2] = canon_atan2f(2.0f * (x5 * x4 + x7 * x6), ((x7 * x7 - x6 * x6) - x5 * x5) + x4 * x4);   /* :488-489 */
}

/* J_next given as a SAMPLE of a device-resident array (grids whose J does not fit the host: C3's 51^6).  Two passes:
 * rec != NULL records, per listed state, the offset of every corner every control touches (cap = nU * 2^D slots per
 * state; values read as 0); keys / vals (sorted offsets + the values the caller gathered there) then serve the real pass. */
typedef struct {
    const int64_t *keys;
    const float *vals;
    int64_t nkeys;
    int64_t *rec;
    int64_t cap;
} sparse_j;

static inline float sparse_get(const sparse_j *sp, int64_t off, int *miss) {
    int64_t lo = 0, hi = sp->nkeys - 1;
    while (lo <= hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sp->keys[mid] == off) return sp->vals[mid];
        if (sp->keys[mid] < off) lo = mid + 1; else hi = mid - 1;
    }
    *miss = 1;
    return 0.f;
}


/* Threads for a loop of `iters` iterations of `work` elementary operations each.  The checker is called once per STAGE by the tests:
 * a 132-state sweep of 151 stages entered 151 parallel regions of as many threads as the host shows (128 on a GPU box whose
 * container may be granted far fewer cores) - 0.12 s per region, 18 s for a problem that takes microseconds.  A thread is worth
 * starting for ~5e5 operations; the count never exceeds what the process may run on (affinity mask, cgroup quota). */
static int usable_cpus(void);
static int threads_for(int64_t iters, int64_t work, int asked) {
    int nt = asked > 0 ? asked : 1;
    const int cap = usable_cpus();
    if (nt > cap) nt = cap;
    const double total = (double)iters * (double)(work > 0 ? work : 1);
    const int64_t by_work = (int64_t)(total / 5.0e5) + 1;
    if ((int64_t)nt > by_work) nt = (int)by_work;
    if ((int64_t)nt > iters) nt = (int)(iters > 0 ? iters : 1);
    return nt < 1 ? 1 : nt;
}

#define DEFINE_BACKUP(T, NAME, FMA)                                                                   \
    static int NAME(const hjb_problem *p, const T *Jn, T *Jout, int32_t *idx_out, int nthreads,        \
                    const int64_t *sel, int64_t nsel, const T *const *jsep, const sparse_j *sp) {      \
        /* sel != NULL: only the listed (whole-grid) states, outputs compact [nsel];                  \
           jsep != NULL: J_next(i) = ((jsep[0][i0] + jsep[1][i1]) + ...) instead of an array;          \
           sp != NULL (with sel): J_next sampled / offsets recorded, see sparse_j */                   \
        const int D = p->D, C = p->C;                                                                  \
        const int tab64 = p->table_dtype == HJB_TAB_F64 && sizeof(T) == 4;                             \
        const int cost64 = p->cost_dtype == HJB_COST_F64 && sizeof(T) == 4;   /* Solver_pos_att.m:800-801 */ \
        T *knots[HJB_MAX_D], *rdx[HJB_MAX_D];                                                          \
        term_t nt[HJB_MAX_D][HJB_MAX_TERMS], ct[HJB_MAX_TERMS];                                        \
        int64_t jstride[HJB_MAX_D];                                                                    \
        for (int a = 0; a < D; ++a) {                                                                  \
            int n = p->n[a];                                                                           \
            knots[a] = (T *)malloc(sizeof(T) * n);                                                     \
            rdx[a] = (T *)malloc(sizeof(T) * n);                                                       \
            for (int i = 0; i < n; ++i) knots[a][i] = (T)p->knots[a][i];                               \
            for (int i = 0; i + 1 < n; ++i) rdx[a][i] = (T)1 / (knots[a][i + 1] - knots[a][i]);        \
            rdx[a][n - 1] = 0;                                                                         \
            for (int k = 0; k < p->n_next_terms[a]; ++k) build_term(p, &p->next_terms[a][k], &nt[a][k]); \
        }                                                                                              \
        for (int k = 0; k < p->n_cost_terms; ++k) build_term(p, &p->cost_terms[k], &ct[k]);            \
        /* slab geometry along the last axis */                                                        \
        int sb = p->slab_begin, se = p->slab_end, hlo = p->halo_lo, hhi = p->halo_hi;                  \
        if (sb == 0 && se == 0) { se = p->n[D - 1]; hlo = hhi = 0; }                                   \
        const int plane0 = sb - hlo;                  /* global plane of local plane 0 */              \
        const int nplanes = (se + hhi) - plane0;                                                       \
        int64_t s = 1, inner = 1;                                                                      \
        for (int a = 0; a < D; ++a) { jstride[a] = s; s *= (a == D - 1) ? nplanes : p->n[a]; }         \
        for (int a = 0; a + 1 < D; ++a) inner *= p->n[a];                                              \
        const int64_t n_owned = inner * (se - sb);                                                     \
        int64_t nU = 1;                                                                                \
        for (int c = 0; c < C; ++c) nU *= p->m[c];                                                     \
        int err = 0;                                                                                   \
        nthreads = threads_for(sel ? nsel : n_owned, nU * ((int64_t)1 << D), nthreads);               \
        _Pragma("omp parallel for schedule(static) num_threads(nthreads)")                            \
        for (int64_t it = 0; it < (sel ? nsel : n_owned); ++it) {                                      \
            const int64_t ls = sel ? sel[it] : it;                                                     \
            int gi[HJB_MAX_G];                                                                         \
            int64_t r = ls;                                                                            \
            for (int a = 0; a < D; ++a) {                                                              \
                int na = (a == D - 1) ? (se - sb) : p->n[a];                                           \
                gi[a] = (int)(r % na);                                                                 \
                r /= na;                                                                               \
            }                                                                                          \
            gi[D - 1] += sb;                                                                           \
            T best = 0;                                                                                \
            int64_t best_label = 0;                                                                    \
            int first = 1;                                                                             \
            for (int c = 0; c < C; ++c) gi[D + c] = 0;                                                 \
            float mq[3] = {0.f, 0.f, 0.f};                                                             \
            if (p->model == HJB_MODEL_QUAT_EULER321)                                                   \
                model_quat_next(p, gi, (float)knots[3][gi[3]], (float)knots[4][gi[4]], (float)knots[5][gi[5]], mq); \
            for (int64_t u = 0; u < nU; ++u) {                                                         \
                T v[1 << HJB_MAX_D];                                                                   \
                T tw[HJB_MAX_D];                                                                       \
                int64_t base = 0;                                                                      \
                for (int a = 0; a < D && tab64; ++a) {       /* Solver_pos_att.m:299-327: double queries */ \
                    double q = 0;                                                                      \
                    for (int k = 0; k < p->n_next_terms[a]; ++k) {                                     \
                        int64_t off = 0;                                                               \
                        for (int d = 0; d < D + C; ++d) off += nt[a][k].stride[d] * gi[d];             \
                        const double x = ((const double *)nt[a][k].data)[off];                         \
                        q = (k == 0) ? x : q + x;                                                      \
                    }                                                                                  \
                    const double *kk = p->knots[a];                                                    \
                    int lo = 0, hi = p->n[a] - 1;                                                      \
                    while (hi - lo > 1) {                                                              \
                        int mid = (lo + hi) >> 1;                                                      \
                        if (kk[mid] <= q) lo = mid; else hi = mid;                                     \
                    }                                                                                  \
                    const double r = 1.0 / (kk[lo + 1] - kk[lo]);                                      \
                    tw[a] = (T)((q - kk[lo]) * r);               /* rounded to the blend's type once */ \
                    int cell = lo;                                                                     \
                    if (a == D - 1) {                                                                  \
                        cell -= plane0;                                                                \
                        if (cell < 0 || cell + 1 >= nplanes) {                                         \
                            err = 1;                                                                   \
                            cell = cell < 0 ? 0 : nplanes - 2;                                         \
                        }                                                                              \
                    }                                                                                  \
                    base += jstride[a] * cell;                                                         \
                }                                                                                      \
                for (int a = 0; a < D && !tab64; ++a) {                                                \
                    T q = (p->model == HJB_MODEL_QUAT_EULER321 && a < 3) ? (T)mq[a] : (T)0;           \
                    for (int k = 0; k < p->n_next_terms[a]; ++k) {                                     \
                        int64_t off = 0;                                                               \
                        for (int d = 0; d < D + C; ++d) off += nt[a][k].stride[d] * gi[d];             \
                        T x = ((const T *)nt[a][k].data)[off];                                         \
                        q = (k == 0) ? x : (T)(q + x);                                                 \
                    }                                                                                  \
                    const T *kk = knots[a];                                                            \
                    int n = p->n[a];                                                                   \
                    /* upper_bound(q) - 1, clamped to [0, n-2] */                                      \
                    int lo = 0, hi = n - 1; /* invariant: answer in [lo, hi-1] */                      \
                    while (hi - lo > 1) {                                                              \
                        int mid = (lo + hi) >> 1;                                                      \
                        if (kk[mid] <= q) lo = mid; else hi = mid;                                     \
                    }                                                                                  \
                    tw[a] = (T)((T)(q - kk[lo]) * rdx[a][lo]);                                         \
                    int cell = lo;                                                                     \
                    if (a == D - 1) {                                                                  \
                        cell -= plane0;                                                                \
                        if (cell < 0 || cell + 1 >= nplanes) {                                         \
                            err = 1;                                                                   \
                            cell = cell < 0 ? 0 : nplanes - 2;                                         \
                        }                                                                              \
                    }                                                                                  \
                    base += jstride[a] * cell;                                                         \
                }                                                                                      \
                for (int c = 0; c < (1 << D); ++c) {                                                   \
                    int64_t off = base;                                                                \
                    for (int a = 0; a < D; ++a)                                                        \
                        if (c & (1 << a)) off += jstride[a];                                           \
                    if (sp && sp->rec) { sp->rec[it * sp->cap + (u << D) + c] = off; v[c] = 0; continue; } \
                    if (sp) { int miss = 0; v[c] = (T)sparse_get(sp, off, &miss); if (miss) err = 2; continue; } \
                    if (!jsep) { v[c] = Jn[off]; continue; }                                           \
                    T sv = 0;                                                                          \
                    for (int a = 0; a < D; ++a) {                                                      \
                        const T x = jsep[a][(off / jstride[a]) % (a == D - 1 ? nplanes : p->n[a])];    \
                        sv = a == 0 ? x : (T)(sv + x);                                                 \
                    }                                                                                  \
                    v[c] = sv;                                                                         \
                }                                                                                      \
                for (int a = 0; a < D; ++a) {                                                          \
                    int half = 1 << (D - 1 - a);                                                       \
                    for (int j = 0; j < half; ++j) v[j] = FMA(tw[a], (T)(v[2 * j + 1] - v[2 * j]), v[2 * j]); \
                }                                                                                      \
                T g = 0;                                                                               \
                if (cost64) {           /* the ordered sum in double, rounded to single ONCE */       \
                    double g64 = 0;                                                                    \
                    for (int k = 0; k < p->n_cost_terms; ++k) {                                        \
                        int64_t off = 0;                                                               \
                        for (int d = 0; d < D + C; ++d) off += ct[k].stride[d] * gi[d];                \
                        const double x = ((const double *)ct[k].data)[off];                            \
                        g64 = (k == 0) ? x : g64 + x;                                                  \
                    }                                                                                  \
                    g = (T)g64;                                                                        \
                }                                                                                      \
                for (int k = 0; k < p->n_cost_terms && !cost64; ++k) {                                 \
                    int64_t off = 0;                                                                   \
                    for (int d = 0; d < D + C; ++d) off += ct[k].stride[d] * gi[d];                    \
                    T x = ((const T *)ct[k].data)[off];                                                \
                    g = (k == 0) ? x : (T)(g + x);                                                     \
                }                                                                                      \
                T tot = (T)(g + v[0]);                                                                 \
                if (first || tot < best) {                                                             \
                    first = 0;                                                                         \
                    best = tot;                                                                        \
                    int64_t lab = 0, mul = 1;                                                          \
                    for (int c = 0; c < C; ++c) { lab += mul * gi[D + c]; mul *= p->m[c]; }            \
                    best_label = lab;                                                                  \
                }                                                                                      \
                /* next control: last control dim fastest (control dim 0 slowest) */                   \
                for (int c = C - 1; c >= 0; --c) {                                                     \
                    if (++gi[D + c] < p->m[c]) break;                                                  \
                    gi[D + c] = 0;                                                                     \
                }                                                                                      \
            }                                                                                          \
            /* owned state -> position in the haloed J layout */                                       \
            int64_t in_plane = ls % inner, pl = ls / inner;                                            \
            Jout[sel ? it : in_plane + inner * (pl + hlo)] = best;                                     \
            if (idx_out) idx_out[sel ? it : ls] = (int32_t)(best_label + p->index_base);               \
        }                                                                                              \
        for (int a = 0; a < D; ++a) { free(knots[a]); free(rdx[a]); }                                  \
        return err == 2 ? HJB_E_INVALID : err ? HJB_E_HALO : HJB_OK;                                   \
    }

DEFINE_BACKUP(float, backup_f32, fmaf)
DEFINE_BACKUP(double, backup_f64, fma)

/* AVX2 + FMA3 form of backup_f32 (the row-vectorised CPU baseline): eight consecutive
 * axis-0 states of one grid row per iteration, one lane each.  Lane by lane it performs exactly the operations of the
 * scalar twin above in the same order (IEEE adds / multiplies, vfmadd for the lerps, the same binary search, strict
 * '<'), so its results are bit-identical - tests/test_oracle_golden.py holds it to that.  float32 arithmetic only
 * (float32 / float16 storage), no state model, J < 2^31 elements; anything else returns HJB_E_UNSUPPORTED.
 * table_dtype HJB_TAB_F64 (Solver_pos_att.m:299-327, the typing bench.py's GPU line runs): the next-state terms are double,
 * a query is summed, located and weighted in double on the double knots (two 4-lane halves), the weight rounded to float
 * once - the scalar twin's tab64 branch, lane by lane. */
/* upper_bound(q) - 1 clamped to [0, n-2] on double knots for four lanes; returns the cell as 64-bit lanes */
static inline __m256i avx2_cell_pd(const double *kk, int n, __m256d q) {
    __m256i lo = _mm256_setzero_si256(), hi = _mm256_set1_epi64x(n - 1);
    for (;;) {
        const __m256i act = _mm256_cmpgt_epi64(_mm256_sub_epi64(hi, lo), _mm256_set1_epi64x(1));
        if (!_mm256_movemask_epi8(act)) break;
        const __m256i mid = _mm256_srli_epi64(_mm256_add_epi64(lo, hi), 1);
        const __m256i le = _mm256_castpd_si256(_mm256_cmp_pd(_mm256_i64gather_pd(kk, mid, 8), q, _CMP_LE_OQ));
        lo = _mm256_blendv_epi8(lo, mid, _mm256_and_si256(act, le));
        hi = _mm256_blendv_epi8(hi, mid, _mm256_andnot_si256(le, act));
    }
    return lo;
}
static int backup_f32_avx2(const hjb_problem *p, const float *Jn, float *Jout, int32_t *idx_out, int nthreads) {
    const int D = p->D, C = p->C;
    if (p->model) return HJB_E_UNSUPPORTED;
    const int tab64 = p->table_dtype == HJB_TAB_F64;
    const int cost64 = p->cost_dtype == HJB_COST_F64;
    float *knots[HJB_MAX_D], *rdx[HJB_MAX_D];
    term_t nt[HJB_MAX_D][HJB_MAX_TERMS], ct[HJB_MAX_TERMS];
    int32_t jstride[HJB_MAX_D];
    int sb = p->slab_begin, se = p->slab_end, hlo = p->halo_lo, hhi = p->halo_hi;
    if (sb == 0 && se == 0) { se = p->n[D - 1]; hlo = hhi = 0; }
    const int plane0 = sb - hlo;
    const int nplanes = (se + hhi) - plane0;
    int64_t s = 1, inner = 1;
    for (int a = 0; a < D; ++a) { jstride[a] = (int32_t)s; s *= (a == D - 1) ? nplanes : p->n[a]; }
    if (s >= ((int64_t)1 << 31)) return HJB_E_UNSUPPORTED;
    for (int a = 0; a + 1 < D; ++a) inner *= p->n[a];
    for (int a = 0; a < D; ++a) {
        int n = p->n[a];
        knots[a] = (float *)malloc(sizeof(float) * n);
        rdx[a] = (float *)malloc(sizeof(float) * n);
        for (int i = 0; i < n; ++i) knots[a][i] = (float)p->knots[a][i];
        for (int i = 0; i + 1 < n; ++i) rdx[a][i] = (float)1 / (knots[a][i + 1] - knots[a][i]);
        rdx[a][n - 1] = 0;
        for (int k = 0; k < p->n_next_terms[a]; ++k) build_term(p, &p->next_terms[a][k], &nt[a][k]);
    }
    for (int k = 0; k < p->n_cost_terms; ++k) build_term(p, &p->cost_terms[k], &ct[k]);
    const int n0 = D > 1 ? p->n[0] : (se - sb);            /* D == 1: the only axis is the sharded one */
    const int64_t n_owned = inner * (se - sb), rows = n_owned / n0;
    int64_t nU = 1;
    for (int c = 0; c < C; ++c) nU *= p->m[c];
    int32_t corner[1 << HJB_MAX_D];
    for (int c = 0; c < (1 << D); ++c) {
        int32_t off = 0;
        for (int a = 0; a < D; ++a)
            if (c & (1 << a)) off += jstride[a];
        corner[c] = off;
    }
    int err = 0;
    nthreads = threads_for(rows, (int64_t)n0 * nU * ((int64_t)1 << D), nthreads);
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t row = 0; row < rows; ++row) {
        int gi[HJB_MAX_G];
        {
            int64_t r = row;
            gi[0] = 0;
            for (int a = 1; a < D; ++a) {
                int na = (a == D - 1) ? (se - sb) : p->n[a];
                gi[a] = (int)(r % na);
                r /= na;
            }
            if (D > 1) gi[D - 1] += sb;
        }
        const __m256i iota = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
        for (int i0 = 0; i0 < n0; i0 += 8) {
            const int nl = n0 - i0 < 8 ? n0 - i0 : 8;       /* lanes past the row's end repeat its last state */
            __m256i li = _mm256_min_epi32(_mm256_add_epi32(_mm256_set1_epi32(i0), iota), _mm256_set1_epi32(n0 - 1));
            if (D == 1) li = _mm256_add_epi32(li, _mm256_set1_epi32(sb));
            __m256 best = _mm256_setzero_ps();
            __m256i lab = _mm256_setzero_si256();
            int first = 1;
            for (int c = 0; c < C; ++c) gi[D + c] = 0;
            for (int64_t u = 0; u < nU; ++u) {
                __m256 v[1 << HJB_MAX_D], tw[HJB_MAX_D];
                __m256i base = _mm256_setzero_si256();
                for (int a = 0; a < D && tab64; ++a) {       /* double queries, weight rounded once */
                    __m256d qd[2] = {_mm256_setzero_pd(), _mm256_setzero_pd()};
                    for (int k = 0; k < p->n_next_terms[a]; ++k) {
                        int64_t off = 0;
                        for (int d = 1; d < D + C; ++d) off += nt[a][k].stride[d] * gi[d];
                        const double *dp = (const double *)nt[a][k].data + off;
                        const int s0 = (int)nt[a][k].stride[0];
                        const __m256i ix = _mm256_mullo_epi32(li, _mm256_set1_epi32(s0));
                        for (int hf = 0; hf < 2; ++hf) {
                            const __m128i ih = hf ? _mm256_extracti128_si256(ix, 1) : _mm256_castsi256_si128(ix);
                            const __m256d x = s0 == 0 ? _mm256_set1_pd(dp[0]) : _mm256_i32gather_pd(dp, ih, 8);
                            qd[hf] = (k == 0) ? x : _mm256_add_pd(qd[hf], x);
                        }
                    }
                    const double *kd = p->knots[a];
                    __m128 th[2];
                    __m128i ch[2];
                    for (int hf = 0; hf < 2; ++hf) {
                        const __m256i lo = avx2_cell_pd(kd, p->n[a], qd[hf]);
                        const __m256d k0 = _mm256_i64gather_pd(kd, lo, 8);
                        const __m256d k1 = _mm256_i64gather_pd(kd, _mm256_add_epi64(lo, _mm256_set1_epi64x(1)), 8);
                        const __m256d r = _mm256_div_pd(_mm256_set1_pd(1.0), _mm256_sub_pd(k1, k0));
                        th[hf] = _mm256_cvtpd_ps(_mm256_mul_pd(_mm256_sub_pd(qd[hf], k0), r));
                        ch[hf] = _mm256_castsi256_si128(_mm256_permutevar8x32_epi32(lo, _mm256_setr_epi32(0, 2, 4, 6, 0, 2, 4, 6)));
                    }
                    tw[a] = _mm256_insertf128_ps(_mm256_castps128_ps256(th[0]), th[1], 1);
                    __m256i cell = _mm256_inserti128_si256(_mm256_castsi128_si256(ch[0]), ch[1], 1);
                    if (a == D - 1) {
                        cell = _mm256_sub_epi32(cell, _mm256_set1_epi32(plane0));
                        __m256i bad = _mm256_or_si256(_mm256_cmpgt_epi32(_mm256_setzero_si256(), cell),
                                                      _mm256_cmpgt_epi32(_mm256_add_epi32(cell, _mm256_set1_epi32(2)), _mm256_set1_epi32(nplanes)));
                        if (_mm256_movemask_epi8(bad)) {
                            err = 1;
                            cell = _mm256_max_epi32(_mm256_min_epi32(cell, _mm256_set1_epi32(nplanes - 2)), _mm256_setzero_si256());
                        }
                    }
                    base = _mm256_add_epi32(base, _mm256_mullo_epi32(cell, _mm256_set1_epi32(jstride[a])));
                }
                for (int a = 0; a < D && !tab64; ++a) {
                    __m256 q = _mm256_setzero_ps();
                    for (int k = 0; k < p->n_next_terms[a]; ++k) {
                        int64_t off = 0;
                        for (int d = 1; d < D + C; ++d) off += nt[a][k].stride[d] * gi[d];
                        const float *dp = (const float *)nt[a][k].data + off;
                        const int s0 = (int)nt[a][k].stride[0];
                        __m256 x = s0 == 0 ? _mm256_set1_ps(dp[0])
                                           : _mm256_i32gather_ps(dp, _mm256_mullo_epi32(li, _mm256_set1_epi32(s0)), 4);
                        q = (k == 0) ? x : _mm256_add_ps(q, x);
                    }
                    const float *kk = knots[a];
                    __m256i lo = _mm256_setzero_si256(), hi = _mm256_set1_epi32(p->n[a] - 1);
                    for (;;) {                               /* upper_bound(q) - 1, clamped to [0, n-2] */
                        __m256i act = _mm256_cmpgt_epi32(_mm256_sub_epi32(hi, lo), _mm256_set1_epi32(1));
                        if (!_mm256_movemask_epi8(act)) break;
                        __m256i mid = _mm256_srai_epi32(_mm256_add_epi32(lo, hi), 1);
                        __m256i le = _mm256_castps_si256(_mm256_cmp_ps(_mm256_i32gather_ps(kk, mid, 4), q, _CMP_LE_OQ));
                        lo = _mm256_blendv_epi8(lo, mid, _mm256_and_si256(act, le));
                        hi = _mm256_blendv_epi8(hi, mid, _mm256_andnot_si256(le, act));
                    }
                    tw[a] = _mm256_mul_ps(_mm256_sub_ps(q, _mm256_i32gather_ps(kk, lo, 4)), _mm256_i32gather_ps(rdx[a], lo, 4));
                    __m256i cell = lo;
                    if (a == D - 1) {
                        cell = _mm256_sub_epi32(cell, _mm256_set1_epi32(plane0));
                        __m256i bad = _mm256_or_si256(_mm256_cmpgt_epi32(_mm256_setzero_si256(), cell),
                                                      _mm256_cmpgt_epi32(_mm256_add_epi32(cell, _mm256_set1_epi32(2)), _mm256_set1_epi32(nplanes)));
                        if (_mm256_movemask_epi8(bad)) {
                            err = 1;
                            cell = _mm256_max_epi32(_mm256_min_epi32(cell, _mm256_set1_epi32(nplanes - 2)), _mm256_setzero_si256());
                        }
                    }
                    base = _mm256_add_epi32(base, _mm256_mullo_epi32(cell, _mm256_set1_epi32(jstride[a])));
                }
                for (int c = 0; c < (1 << D); ++c) v[c] = _mm256_i32gather_ps(Jn + corner[c], base, 4);
                for (int a = 0; a < D; ++a) {
                    int half = 1 << (D - 1 - a);
                    for (int j = 0; j < half; ++j) v[j] = _mm256_fmadd_ps(tw[a], _mm256_sub_ps(v[2 * j + 1], v[2 * j]), v[2 * j]);
                }
                __m256 g = _mm256_setzero_ps();
                if (cost64) {           /* the ordered sum in double (two 4-lane halves), rounded to single once */
                    __m256d gd[2] = {_mm256_setzero_pd(), _mm256_setzero_pd()};
                    for (int k = 0; k < p->n_cost_terms; ++k) {
                        int64_t off = 0;
                        for (int d = 1; d < D + C; ++d) off += ct[k].stride[d] * gi[d];
                        const double *dp = (const double *)ct[k].data + off;
                        const int s0 = (int)ct[k].stride[0];
                        const __m256i ix = _mm256_mullo_epi32(li, _mm256_set1_epi32(s0));
                        for (int hf = 0; hf < 2; ++hf) {
                            const __m128i ih = hf ? _mm256_extracti128_si256(ix, 1) : _mm256_castsi256_si128(ix);
                            const __m256d x = s0 == 0 ? _mm256_set1_pd(dp[0]) : _mm256_i32gather_pd(dp, ih, 8);
                            gd[hf] = (k == 0) ? x : _mm256_add_pd(gd[hf], x);
                        }
                    }
                    g = _mm256_insertf128_ps(_mm256_castps128_ps256(_mm256_cvtpd_ps(gd[0])), _mm256_cvtpd_ps(gd[1]), 1);
                }
                for (int k = 0; k < p->n_cost_terms && !cost64; ++k) {
                    int64_t off = 0;
                    for (int d = 1; d < D + C; ++d) off += ct[k].stride[d] * gi[d];
                    const float *dp = (const float *)ct[k].data + off;
                    const int s0 = (int)ct[k].stride[0];
                    __m256 x = s0 == 0 ? _mm256_set1_ps(dp[0])
                                       : _mm256_i32gather_ps(dp, _mm256_mullo_epi32(li, _mm256_set1_epi32(s0)), 4);
                    g = (k == 0) ? x : _mm256_add_ps(g, x);
                }
                const __m256 tot = _mm256_add_ps(g, v[0]);
                int64_t label = 0, mul = 1;
                for (int c = 0; c < C; ++c) { label += mul * gi[D + c]; mul *= p->m[c]; }
                const __m256 take = first ? _mm256_castsi256_ps(_mm256_set1_epi32(-1)) : _mm256_cmp_ps(tot, best, _CMP_LT_OQ);
                first = 0;
                best = _mm256_blendv_ps(best, tot, take);
                lab = _mm256_blendv_epi8(lab, _mm256_set1_epi32((int32_t)(label + p->index_base)), _mm256_castps_si256(take));
                for (int c = C - 1; c >= 0; --c) {
                    if (++gi[D + c] < p->m[c]) break;
                    gi[D + c] = 0;
                }
            }
            const int64_t ls = row * n0 + i0;                /* first owned state of this vector */
            const int64_t in_plane = ls % inner, pl = ls / inner;
            const __m256i keep = _mm256_cmpgt_epi32(_mm256_set1_epi32(nl), iota);
            _mm256_maskstore_ps(Jout + in_plane + inner * (pl + hlo), keep, best);
            if (idx_out) _mm256_maskstore_epi32(idx_out + ls, keep, lab);
        }
    }
    for (int a = 0; a < D; ++a) { free(knots[a]); free(rdx[a]); }
    return err ? HJB_E_HALO : HJB_OK;
}

/* one backup.  J buffers are in the haloed slab layout of hjb_problem (whole
 * grid when the slab fields are zero); idx_out covers owned states only. */
/* IEEE binary16 <-> float (F16C): widening is exact, narrowing rounds to nearest even - the same
 * conversions the GPU performs for HJB_F16S storage. */
static float h2f(uint16_t h) { return _cvtsh_ss(h); }
static uint16_t f2h(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }

static int64_t j_elems_of(const hjb_problem *p) {
    int D = p->D, sb = p->slab_begin, se = p->slab_end, hlo = p->halo_lo, hhi = p->halo_hi;
    if (sb == 0 && se == 0) { se = p->n[D - 1]; hlo = hhi = 0; }
    int64_t s = (se + hhi) - (sb - hlo);
    for (int a = 0; a + 1 < D; ++a) s *= p->n[a];
    return s;
}

int orc_backup_stage(const hjb_problem *p, const void *J_next, void *J_out, int32_t *idx_out, int nthreads) {
    int st = validate(p);
    if (st) return st;
    if (p->dtype == HJB_F16S) {   /* float32 arithmetic on widened J, result narrowed on store */
        int64_t ne = j_elems_of(p);
        float *a = (float *)malloc(sizeof(float) * (ne > 0 ? ne : 1)), *b = (float *)malloc(sizeof(float) * (ne > 0 ? ne : 1));
        if (!a || !b) { free(a); free(b); return HJB_E_NOMEM; }
        for (int64_t i = 0; i < ne; ++i) { a[i] = h2f(((const uint16_t *)J_next)[i]); b[i] = h2f(((const uint16_t *)J_out)[i]); }
        st = backup_f32(p, a, b, idx_out, nthreads, NULL, 0, NULL, NULL);
        for (int64_t i = 0; i < ne; ++i) ((uint16_t *)J_out)[i] = f2h(b[i]);
        free(a); free(b);
        return st;
    }
    if (p->dtype == HJB_F32) return backup_f32(p, (const float *)J_next, (float *)J_out, idx_out, nthreads, NULL, 0, NULL, NULL);
    return backup_f64(p, (const double *)J_next, (double *)J_out, idx_out, nthreads, NULL, 0, NULL, NULL);
}

/* the same backup by the AVX2 row-vectorised form (float32 arithmetic only) */
int orc_backup_stage_avx2(const hjb_problem *p, const void *J_next, void *J_out, int32_t *idx_out, int nthreads) {
    int st = validate(p);
    if (st) return st;
    if (p->dtype == HJB_F16S) {   /* float32 arithmetic on widened J, result narrowed on store */
        int64_t ne = j_elems_of(p);
        float *a = (float *)malloc(sizeof(float) * (ne > 0 ? ne : 1)), *b = (float *)malloc(sizeof(float) * (ne > 0 ? ne : 1));
        if (!a || !b) { free(a); free(b); return HJB_E_NOMEM; }
        for (int64_t i = 0; i < ne; ++i) { a[i] = h2f(((const uint16_t *)J_next)[i]); b[i] = h2f(((const uint16_t *)J_out)[i]); }
        st = backup_f32_avx2(p, a, b, idx_out, nthreads);
        for (int64_t i = 0; i < ne; ++i) ((uint16_t *)J_out)[i] = f2h(b[i]);
        free(a); free(b);
        return st;
    }
    if (p->dtype == HJB_F32) return backup_f32_avx2(p, (const float *)J_next, (float *)J_out, idx_out, nthreads);
    return HJB_E_UNSUPPORTED;
}

/* Backup of a LIST of states of the whole grid (no slab) with J_next given in separable form
 * J_next(i) = ((jsep[0][i0] + jsep[1][i1]) + ...) - lets the checker sample grids whose J does not fit in
 * host memory (C3: 51^6).  HJB_F32 only.  Outputs are compact: J_out[k], idx_out[k] for states[k]. */
int orc_backup_states(const hjb_problem *p, const float *const *jsep, const int64_t *states, int64_t nstates,
                      float *J_out, int32_t *idx_out, int nthreads) {
    int st = validate(p);
    if (st) return st;
    if (p->dtype != HJB_F32 || p->slab_begin || p->slab_end) return HJB_E_UNSUPPORTED;
    return backup_f32(p, NULL, J_out, idx_out, nthreads, states, nstates, jsep, NULL);
}

/* Backup of a LIST of whole-grid states from a whole-grid J_next array held by the host (a deep-sweep check: the GPU's OWN
 * previous stage, downloaded; 24^6 floats are 764 MB).  HJB_F32, no slab.  Outputs compact. */
int orc_backup_states_from_J(const hjb_problem *p, const float *J_next, const int64_t *states, int64_t nstates,
                             float *J_out, int32_t *idx_out, int nthreads) {
    int st = validate(p);
    if (st) return st;
    if (p->dtype != HJB_F32 || p->slab_begin || p->slab_end) return HJB_E_UNSUPPORTED;
    return backup_f32(p, J_next, J_out, idx_out, nthreads, states, nstates, NULL, NULL);
}

/* The same for a J_next that exists on the device only (C3: 70 GB).  Pass 1 (orc_backup_states_touch): offsets[k * cap + j],
 * cap = nU * 2^D, = every J_next element state k's backup reads (duplicates included).  The caller gathers the unique,
 * sorted offsets from the device (hjb_device_gather) and pass 2 (orc_backup_states_sparse) performs the backup on them;
 * an offset missing from `keys` is an error (HJB_E_INVALID), never a silent zero. */
int orc_backup_states_touch(const hjb_problem *p, const int64_t *states, int64_t nstates, int64_t *offsets, int64_t cap,
                            int nthreads) {
    int st = validate(p);
    if (st) return st;
    if (p->dtype != HJB_F32 || p->slab_begin || p->slab_end) return HJB_E_UNSUPPORTED;
    int64_t nU = 1;
    for (int c = 0; c < p->C; ++c) nU *= p->m[c];
    if (cap != (nU << p->D)) return HJB_E_INVALID;
    float *J = (float *)malloc(sizeof(float) * (size_t)(nstates > 0 ? nstates : 1));
    if (!J) return HJB_E_NOMEM;
    sparse_j sp = {NULL, NULL, 0, offsets, cap};
    st = backup_f32(p, NULL, J, NULL, nthreads, states, nstates, NULL, &sp);
    free(J);
    return st;
}

int orc_backup_states_sparse(const hjb_problem *p, const int64_t *keys, const float *vals, int64_t nkeys,
                             const int64_t *states, int64_t nstates, float *J_out, int32_t *idx_out, int nthreads) {
    int st = validate(p);
    if (st) return st;
    if (p->dtype != HJB_F32 || p->slab_begin || p->slab_end) return HJB_E_UNSUPPORTED;
    sparse_j sp = {keys, vals, nkeys, NULL, 0};
    return backup_f32(p, NULL, J_out, idx_out, nthreads, states, nstates, NULL, &sp);
}

/* hjb_solve_opts.monitor_single: MATLAB's sum(F_gI.Values(:)) of a single array (Solver_pos_att.m:274) is a
 * single-precision sum in an order MathWorks does not document.  The library states ITS order (csrc/kernels_reduce.h)
 * and this is that order, element for element, in float32:
 *   accumulator a (0 <= a < 131072) takes elements a, a + 131072, ... in ascending order;
 *   block b = a / 256 reduces its 256 accumulators by pairwise halving (s = 128 .. 1: acc[t] += acc[t + s]);
 *   accumulator t (0 <= t < 256) of the final pass adds block sums t, t + 256 in that order; pairwise halving again. */
static float single_sum(int dtype, const void *A, int64_t n) {
    enum { NB = 512, NT = 256 };
    float *acc = (float *)calloc((size_t)NB * NT, sizeof(float));
    float part[NB], fin[NT];
    for (int64_t a = 0; a < (int64_t)NB * NT; ++a) {
        float s = 0.0f;
        for (int64_t i = a; i < n; i += (int64_t)NB * NT) {
            const float x = dtype == HJB_F16S ? h2f(((const uint16_t *)A)[i]) : ((const float *)A)[i];
            s = s + x;
        }
        acc[a] = s;
    }
    for (int b = 0; b < NB; ++b) {
        float *v = acc + (size_t)b * NT;
        for (int s = NT / 2; s > 0; s >>= 1)
            for (int t = 0; t < s; ++t) v[t] = v[t] + v[t + s];
        part[b] = v[0];
    }
    for (int t = 0; t < NT; ++t) {
        float s = 0.0f;
        for (int b = t; b < NB; b += NT) s = s + part[b];
        fin[t] = s;
    }
    for (int s = NT / 2; s > 0; s >>= 1)
        for (int t = 0; t < s; ++t) fin[t] = fin[t] + fin[t + s];
    free(acc);
    return fin[0];
}

/* The monitor's two sums (Solver_pos_att.m:274-275) of a given J / label array, exactly as orc_sweep forms them:
 * out[0] = sum(J) in float64 (single != 0 and a float32 / binary16 J: the stated float32 tree, widened), out[1] = the
 * exact label sum.  For tests that check the library's monitor on a J the GPU produced (grids too big to sweep here). */
int orc_monitor_sums(int dtype, const void *J, const int32_t *idx, int64_t n, int single, double *out2) {
    double fs = 0, is = 0;
    for (int64_t i = 0; i < n; ++i) {
        fs += dtype == HJB_F16S ? (double)h2f(((const uint16_t *)J)[i])
                                : (dtype == HJB_F32 ? (double)((const float *)J)[i] : ((const double *)J)[i]);
        if (idx) is += (double)idx[i];
    }
    if (single && dtype != HJB_F64) fs = (double)single_sum(dtype, J, n);
    out2[0] = fs; out2[1] = is;
    return 0;
}

/* whole-grid backward sweep with the same outputs as hjb_solve. */
int orc_sweep(const hjb_problem *p, const hjb_solve_opts *o, hjb_result *res, int nthreads) {
    int st = validate(p);
    if (st) return st;
    if (p->slab_begin || p->slab_end) return HJB_E_UNSUPPORTED;
    int64_t nS = 1;
    for (int a = 0; a < p->D; ++a) nS *= p->n[a];
    size_t es = p->dtype == HJB_F16S ? 2 : (p->dtype == HJB_F32 ? 4 : 8);
    char *A = (char *)calloc(nS, es), *B = (char *)calloc(nS, es);
    int32_t *idx = (int32_t *)calloc(nS, sizeof(int32_t));
    if (!A || !B || !idx) return HJB_E_NOMEM;
    if (o->terminal) memcpy(A, o->terminal, nS * es);
    double fprev = 0, iprev = 0;
    int done = 0, early = 0;
    double e = 0, e2 = 0;
    for (int k_s = o->n_stages; k_s >= 1; --k_s) {
        st = orc_backup_stage(p, A, B, idx, nthreads);
        if (st) break;
        char *t = A; A = B; B = t;
        ++done;
        if (o->J_stages) memcpy((char *)o->J_stages + (size_t)(k_s - 1) * nS * es, A, nS * es);
        if (o->idx_stages) memcpy((int32_t *)o->idx_stages + (size_t)(k_s - 1) * nS, idx, nS * sizeof(int32_t));
        if (o->monitor_period > 0 && (k_s % o->monitor_period) == 0) {
            double fs = 0, is = 0;
            for (int64_t i = 0; i < nS; ++i) {
                fs += p->dtype == HJB_F16S ? (double)h2f(((uint16_t *)A)[i])
                                           : (p->dtype == HJB_F32 ? (double)((float *)A)[i] : ((double *)A)[i]);
                is += (double)idx[i];
            }
            const int msingle = o->monitor_single && p->dtype != HJB_F64;
            if (msingle) fs = (double)single_sum(p->dtype, A, nS);
            /* Solver_pos_att.m:276-282: single fsum50 => single subtraction, and abs(e) < tol compared in single */
            e = msingle ? (double)((float)fs - (float)fprev) : fs - fprev;
            e2 = is - iprev;
            fprev = fs; iprev = is;
            if (o->progress) o->progress(o->progress_user, k_s, e, e2, 0.0);
            if (msingle ? (fabsf((float)e) < (float)o->monitor_tol) : (fabs(e) < o->monitor_tol)) { early = 1; break; }
        }
    }
    if (o->J_final) memcpy(o->J_final, A, nS * es);
    if (o->idx_final) memcpy(o->idx_final, idx, nS * sizeof(int32_t));
    if (res) { res->stages_done = done; res->stopped_early = early; res->sweep_ms = 0; res->last_e = e; res->last_e2 = e2; }
    free(A); free(B); free(idx);
    return st;
}

/* Batched gridded lookup (policy use: Solver_position.m:144-146 'nearest',
 * Dynamic_Solver.m:132-135 'linear'); same canonical arithmetic as the sweep.
 * method 0 = nearest (upper knot at the midpoint), 1 = linear with extrapolation. */
#define DEFINE_LOOKUP(T, NAME, FMA)                                                                  \
    static void NAME(int D, const int32_t *n, const double *const *knots, const T *V, int64_t nq,      \
                     const T *Q, int method, T *out) {                                                \
        T *kk[HJB_MAX_D], *rdx[HJB_MAX_D];                                                            \
        int64_t stride[HJB_MAX_D], s = 1;                                                             \
        for (int a = 0; a < D; ++a) {                                                                 \
            kk[a] = (T *)malloc(sizeof(T) * n[a]);                                                    \
            rdx[a] = (T *)malloc(sizeof(T) * n[a]);                                                   \
            for (int i = 0; i < n[a]; ++i) kk[a][i] = (T)knots[a][i];                                 \
            for (int i = 0; i + 1 < n[a]; ++i) rdx[a][i] = (T)1 / (kk[a][i + 1] - kk[a][i]);          \
            stride[a] = s;                                                                            \
            s *= n[a];                                                                                \
        }                                                                                             \
        for (int64_t i = 0; i < nq; ++i) {                                                            \
            T tw[HJB_MAX_D], v[1 << HJB_MAX_D];                                                       \
            int64_t base = 0;                                                                         \
            for (int a = 0; a < D; ++a) {                                                             \
                T q = Q[i * D + a];                                                                   \
                int lo = 0, hi = n[a] - 1;                                                            \
                while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (kk[a][mid] <= q) lo = mid; else hi = mid; } \
                if (method == 0) { if ((T)(q - kk[a][lo]) >= (T)(kk[a][lo + 1] - q)) ++lo; tw[a] = 0; } \
                else tw[a] = (T)((T)(q - kk[a][lo]) * rdx[a][lo]);                                    \
                base += stride[a] * lo;                                                               \
            }                                                                                         \
            if (method == 0) { out[i] = V[base]; continue; }                                          \
            for (int c = 0; c < (1 << D); ++c) {                                                      \
                int64_t off = base;                                                                   \
                for (int a = 0; a < D; ++a) if (c & (1 << a)) off += stride[a];                       \
                v[c] = V[off];                                                                        \
            }                                                                                         \
            for (int a = 0; a < D; ++a) {                                                             \
                int half = 1 << (D - 1 - a);                                                          \
                for (int j = 0; j < half; ++j) v[j] = FMA(tw[a], (T)(v[2 * j + 1] - v[2 * j]), v[2 * j]); \
            }                                                                                         \
            out[i] = v[0];                                                                            \
        }                                                                                             \
        for (int a = 0; a < D; ++a) { free(kk[a]); free(rdx[a]); }                                    \
    }
DEFINE_LOOKUP(float, lookup_f32, fmaf)
DEFINE_LOOKUP(double, lookup_f64, fma)

int orc_lookup(int dtype, int D, const int32_t *n, const double *const *knots, const void *V, int64_t nq,
               const void *Q, int method, void *out) {
    if (D < 1 || D > HJB_MAX_D) return HJB_E_UNSUPPORTED;
    if (dtype == HJB_F32) lookup_f32(D, n, knots, (const float *)V, nq, (const float *)Q, method, (float *)out);
    else lookup_f64(D, n, knots, (const double *)V, nq, (const double *)Q, method, (double *)out);
    return HJB_OK;
}

/* CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (v2 cpu.max, v1 cfs_quota_us) */
static int usable_cpus(void) {
    static int cached = 0;
    if (cached) return cached;
    int n = 1;
#ifdef _OPENMP
    n = omp_get_max_threads();
#endif
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a > 0 && a < n) n = a;
    }
    long long quota = -1, period = 100000;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) != NULL) {
        if (fscanf(f, "%lld", &quota) != 1) quota = -1;
        fclose(f);
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) != NULL) {
            if (fscanf(f, "%lld", &period) != 1) period = 100000;
            fclose(f);
        }
    }
    if (quota > 0 && period > 0) {
        const int c = (int)((quota + period - 1) / period);
        if (c > 0 && c < n) n = c;
    }
    cached = n < 1 ? 1 : n;
    return cached;
}

int orc_max_threads(void) { return usable_cpus(); }
