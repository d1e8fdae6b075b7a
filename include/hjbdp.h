/* hjbdp.h - C ABI of libhjbdp: the Bellman-backup hot path of grid-based HJB
 * dynamic programming on AMD MI355X (gfx950).
 *
 * The reference (abdolrezat/Optimal-Control-Dynamic-Programming, pure MATLAB) has
 * no FFI seam; the seam this library replaces is the per-stage statement every
 * solver repeats inside its stage loop
 *
 *     [F.Values, idx] = min( J_stage + F(x_next_1,...,x_next_D), [], ctrl_dim )
 *
 *   test/Dynamic_Solver.m:207-210           (J_state_M, called from run :86-102)
 *   position-control/Solver_position.m:135-137   (simplified_run :132-141)
 *   attitude-control/Solver_attitude.m:239-241   (simplified_run :236-247)
 *   attitude-control/Solver_attitude.m:400-409   (calculate_J_U_opt_state_M, run :280-287)
 *   pos-att/Solver_pos_att.m:272                 (calculate_one_channel_U_Opt :270-286)
 *
 * plus the loop around it (hjb_solve) including the pos-att early-stop monitor
 * (Solver_pos_att.m:268-285).  F is griddedInterpolant(...,'linear'): N-linear
 * interpolation with linear extrapolation; min returns the first minimal index.
 *
 * Conventions (MATLAB's): all arrays COLUMN-MAJOR, first index fastest; grid
 * dims 0..D-1 are state axes, D..D+C-1 control axes; argmin labels enumerate the
 * control dims column-major (ndgrid order) and are index_base-based.
 *
 * Plain C, caller-owned host pointers, no callbacks except the optional
 * progress function; never throws.  A MATLAB host binds this header with
 * loadlibrary/calllib; tests bind it with ctypes.
 */
#ifndef HJBDP_H
#define HJBDP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HJB_MAX_D 6      /* state dims   (Solver_attitude.run: 6)            */
#define HJB_MAX_C 3      /* control dims (Solver_attitude.run: U1,U2,U3)     */
#define HJB_MAX_G 9      /* D + C        (reshape_states: dims 1..9)         */
#define HJB_MAX_TERMS 12 /* broadcast terms per quantity (attitude cost: 9)      */

/* status codes */
#define HJB_OK 0
#define HJB_E_INVALID 1      /* bad argument / inconsistent problem           */
#define HJB_E_UNSUPPORTED 2  /* valid but not supported (D, C, dtype)         */
#define HJB_E_DEVICE 3       /* HIP error; text via hjb_last_error            */
#define HJB_E_NOMEM 4
#define HJB_E_HALO 5         /* a query left the slab's halo (multi-GPU)      */

#define HJB_F32 0
#define HJB_F64 1
#define HJB_F16S 2   /* float32 arithmetic, tables and knots; J buffers (terminal, J_final, J_stages,
                        device J) stored as IEEE binary16, rounded to nearest even on store -
                        halves the cost-to-go footprint (288 GB sizing of the 6-D grids)   */

/* One broadcast term of an ordered sum  q = ((t0 + t1) + t2) + ...
 * This is MATLAB implicit expansion of vectors/arrays reshaped onto the
 * D+C grid dims (Solver_attitude.m:717-742 reshape_states; Solver_pos_att.m
 * :307-314, :791-801): bit d of mask set <=> the term varies along grid dim d.
 * data: host array of the problem's dtype, column-major over the masked dims
 * in increasing dim order. */
typedef struct hjb_term {
    uint32_t mask;
    uint32_t reserved;
    const void *data;
} hjb_term;

typedef struct hjb_problem {
    int32_t D;                      /* number of state axes, 1..HJB_MAX_D      */
    int32_t C;                      /* number of control axes, 1..HJB_MAX_C    */
    int32_t n[HJB_MAX_D];           /* state grid sizes (>= 2)                 */
    int32_t m[HJB_MAX_C];           /* control grid sizes (>= 1)               */
    int32_t dtype;                  /* HJB_F32 / HJB_F64 / HJB_F16S: arithmetic type of J,
                                       tables, knots and weights               */
    int32_t index_base;             /* 0 or 1 (MATLAB) for argmin labels       */
    const double *knots[HJB_MAX_D]; /* grid vectors, strictly increasing, may be
                                       non-uniform (Solver_pos_att.m:906-918);
                                       rounded to dtype inside                 */
    /* x_next_a = ordered sum of terms (a_D_M Dynamic_Solver.m:184-188;
       next_stage_states_simplified Solver_pos_att.m:299-328; ...)             */
    int32_t n_next_terms[HJB_MAX_D];
    hjb_term next_terms[HJB_MAX_D][HJB_MAX_TERMS];
    /* stage cost g = ordered sum of terms (g_D Dynamic_Solver.m:196-200;
       J_current_reshaped Solver_pos_att.m:784-802; ...)                       */
    int32_t n_cost_terms;
    int32_t idx_dtype;              /* storage type of the argmin labels (idx_final, idx_stages, d_idx_out):
                                       HJB_IDX_I32 (0, default), HJB_IDX_U8, HJB_IDX_U16, or HJB_IDX_AUTO = the narrowest
                                       that holds nU - 1 + index_base (MATLAB's U_Optimal_id of Solver_pos_att.m:272
                                       carries 9 distinct values: one byte per state instead of four); the width in
                                       effect is hjb_info.idx_bytes */
    hjb_term cost_terms[HJB_MAX_TERMS];
    /* Slab decomposition along the LAST state axis (multi-GPU; all four zero =
       whole grid).  The handle owns planes [slab_begin, slab_end) and every J
       buffer it is given covers planes [slab_begin-halo_lo, slab_end+halo_hi). */
    int32_t slab_begin, slab_end, halo_lo, halo_hi;
    /* Optional on-the-fly model of the LEADING state axes (those whose next value does not depend on the
       control).  HJB_MODEL_NONE: every axis is described by next_terms.  HJB_MODEL_QUAT_EULER321 (D == 6, C == 3):
       axes 0,1,2 = (yaw, pitch, roll), axes 3,4,5 = (w1, w2, w3); n_next_terms[0..2] must be 0 and the next angles
       are computed per state as attitude-control/Solver_attitude.m:449-489 does (Euler step of the quaternion
       kinematics with step model_h, renormalise, back to Euler angles) from model_tables[0..3] = the quaternion
       components x4,x5,x6,x7 of the grid angles (Solver_attitude.m:419-421; arrays over (n[0],n[1],n[2]),
       column-major, problem dtype HJB_F32/HJB_F16S) and the knots of axes 3..5.  This replaces the three
       nS-sized next-angle tables, which do not fit at 51^6.  atan2/asin use the library's own
       fixed polynomial forms, so that results are reproducible bit for bit across CPUs/GPUs. */
    int32_t model;
    int32_t table_dtype;            /* HJB_TAB_DEFAULT (0): next-state terms are arrays of the problem dtype and the
                                       queries are formed, located and weighted in it.  HJB_TAB_F64 (dtype HJB_F32 /
                                       HJB_F16S only): the data of every next_terms entry are float64 - the reference's
                                       pos-att typing (Solver_pos_att.m:299-327 keeps x_next .. w_next in double while
                                       F_gI.Values is single, :264-265): each query is summed in double, its cell is found
                                       on the float64 knots, its weight (q - k[c]) * (1 / (k[c+1] - k[c])) is formed in
                                       double and rounded to float32 ONCE; the blend and the cost stay float32.  The
                                       (cell, weight) tables are stage-invariant, so this costs nothing per stage; only
                                       the table-driven stage kernels (variants 5, 6, 7 and the multi-stage 2-D kernel)
                                       serve such a problem.  cost_terms stay arrays of the problem dtype. */
    double model_h;
    const void *model_tables[4];
    int32_t cost_dtype;             /* HJB_COST_DEFAULT (0): cost_terms are arrays of the problem dtype, summed in it.
                                       HJB_COST_F64 (dtype HJB_F32 / HJB_F16S, no state model): the data of every
                                       cost_terms entry are float64 and the stage cost of a (state, control) is the ordered
                                       sum of the terms IN DOUBLE, rounded to float32 ONCE - the reference's
                                       J_current_M = single(Qx*x.^2 + Qv*v.^2 + Qw*w.^2 + Qt*t.^2 + (R*f1.^2 + ...))
                                       (Solver_pos_att.m:800-801: a double expression under implicit expansion, one cast)
                                       without the [n_x,n_v,n_t,n_w,nU] array: bit-identical to passing that array as one
                                       term, at any grid size.  The state part is summed once per state, each control adds
                                       its part in double and rounds.  Served by the table-driven kernel (variant 5) and
                                       the column sweep (variant 7, state terms + one control term); ~10 % slower than
                                       float32 terms on C4. */
    int32_t reserved_;
} hjb_problem;

#define HJB_IDX_I32 0
#define HJB_IDX_U8 1
#define HJB_IDX_U16 2
#define HJB_IDX_AUTO 3

#define HJB_TAB_DEFAULT 0
#define HJB_TAB_F64 1

#define HJB_COST_DEFAULT 0
#define HJB_COST_F64 1

#define HJB_MODEL_NONE 0
#define HJB_MODEL_QUAT_EULER321 1

typedef struct hjb_handle_s *hjb_handle;

/* optional progress callback, replaces fprintf/waitbar in the stage loops
 * (Dynamic_Solver.m:101, Solver_pos_att.m:278): called at monitor points (and, with
 * hjb_solve_opts.progress_every_stage, after every stage) with the reference's k_s,
 * e = d(sum J), e2 = d(sum idx), elapsed seconds. */
typedef void (*hjb_progress_fn)(void *user, int32_t k_s, double e, double e2, double seconds);

/* Optional probe block = the reference's debug taps (test/Dynamic_Solver.m:212-219, `checkstagesXJF`): per stage the
 * reference copies the fixed sub-block (50:55, 52:57, 105) of J_current_state, X_next_M1, X_next_M2 (and, commented
 * out, of J_F_next) into *_check(:,:,k).  Here: any rectangular sub-block of states [lo, hi) (0-based, half open -
 * 50:55 is lo 49, hi 55) at ONE control (0-based index per control dim - 105 is 104).  With B = prod(hi - lo) block
 * states in column-major order, plane k_s - 1 of each output holds stage k_s (as J_stages does):
 *   g        [B, n_stages]     the stage cost g(x, u)                      (J_current_state_check)
 *   x_next   [B, D, n_stages]  the next-state coordinate of every axis     (X_next_M1_check, X_next_M2_check, ...)
 *   j_interp [B, n_stages]     J_{k+1} interpolated at x_next              (J_F_next_check)
 * all in the problem's arithmetic dtype (float for HJB_F16S); any of the three may be NULL. */
typedef struct hjb_probe {
    int32_t lo[HJB_MAX_D], hi[HJB_MAX_D];
    int32_t control[HJB_MAX_C];
    int32_t reserved;
    void *g;
    void *x_next;
    void *j_interp;
} hjb_probe;

typedef struct hjb_solve_opts {
    int32_t n_stages;        /* number of backups: N-1 (Dynamic_Solver.m:86), N_stage-1 */
    int32_t monitor_period;  /* 0 = off; 50 in Solver_pos_att.m:273                     */
    double monitor_tol;      /* 1e-2 in Solver_pos_att.m:269.  By default the two sums are float64 sums (fixed
                                reduction tree, reproducible); MATLAB's sum() of the single array F.Values is a
                                single-precision sum, so |e| < tol can first hold at a different monitor point there:
                                see monitor_single below */
    const void *terminal;    /* J_N [nS] dtype, NULL = zeros (Dynamic_Solver.m:83-84)   */
    void *J_final;           /* out [nS] dtype: J of the last computed stage (may be NULL) */
    void *idx_final;         /* out [nS] labels of hjb_problem.idx_dtype (int32 by default): argmin of the last
                                computed stage                                          */
    void *J_stages;          /* out [nS * n_stages] or NULL: stage with reference index
                                k_s (1-based, counting down from n_stages) is written
                                to plane k_s-1  (test_coder.m:32 J_star(:,:,k_s))       */
    void *idx_stages;        /* out [nS * n_stages] labels or NULL (Dynamic_Solver.m:100) */
    hjb_progress_fn progress;
    void *progress_user;
    const hjb_probe *probe;  /* NULL = no debug taps (Dynamic_Solver.m:212-219)                    */
    int32_t progress_every_stage; /* 0: progress is called at monitor points only; 1: after every stage (the reference
                                     prints per stage, Dynamic_Solver.m:101; e, e2 are then 0 between monitor points) */
    int32_t monitor_single;  /* 1: the monitor's sum of J is accumulated in float32, as MATLAB's sum() of the single
                                array F_gI.Values is (Solver_pos_att.m:274; dtype HJB_F32 / HJB_F16S, one device).
                                MATLAB's own summation order is not documented; the order used here is the fixed
                                tree stated in csrc/kernels_reduce.h, restated by the oracle.  The label sum stays
                                exact (U_Optimal_id is a double array there).  The difference e = fsum50 - fsum50_prev
                                and the test abs(e) < tol are then single-precision too (:276-282; MATLAB casts the
                                double tol to single).  0: float64 sums, float64 difference and test */
} hjb_solve_opts;

typedef struct hjb_result {
    int32_t stages_done;     /* < n_stages if the monitor stopped the sweep             */
    int32_t stopped_early;
    double sweep_ms;         /* device time of the stage loop (HIP events)              */
    double last_e, last_e2;  /* monitor deltas at the last monitor point                */
} hjb_result;

typedef struct hjb_info {
    int64_t n_states;        /* states this handle owns                                 */
    int64_t n_controls;
    int64_t j_elems;         /* elements of a J buffer (owned + halo planes)            */
    int32_t kernel_variant;  /* which stage kernel hjb_create selected                  */
    int32_t lds_bytes;
    int32_t block, grid;
    int32_t halo_needed_lo;  /* conservative halo the tables imply (planes)             */
    int32_t halo_needed_hi;
    int32_t idx_bytes;       /* bytes per stored argmin label: 4, 1 or 2 (hjb_problem.idx_dtype resolved) */
    int32_t table_dtype;     /* HJB_TAB_* in effect                                     */
    int32_t cost_dtype;      /* HJB_COST_* in effect                                    */
    int32_t reserved_;
} hjb_info;

const char *hjb_version(void);
const char *hjb_status_string(int32_t status);
/* number of visible HIP devices, or 0 */
int32_t hjb_device_count(void);
/* Fault injection for the test suite (NOT for production hosts): an explicit in-process call - the library never reads
 * the environment to change behaviour.  Keys: "fail_tab64_scratch" (value != 0: the float64 table build's scratch
 * allocation fails, so that hjb_create's refusal of an unservable HJB_TAB_F64 problem can be tested),
 * "fail_tabled_alloc" (the (cell, t) table allocation of the table-driven kernels fails: HJB_COST_F64's refusal),
 * "rccl_only_env" (the RCCL loader tries $HJBDP_RCCL_LIB only: a host without librccl). */
int32_t hjb_test_hook(const char *key, int64_t value);

/* Threading: a handle is not thread-safe - drive each handle from one host thread.  DIFFERENT handles may be
 * driven from different threads at the same time (each hjb_solve runs on its handle's own HIP stream): this is how
 * the independent channels of the spacecraft solvers run side by side.  The library serialises only what HIP
 * cannot overlap safely (stream capture against allocation / synchronous copies); hjb_create and hjb_solve wait for
 * their own set-up (issued on the null stream), never for the whole device, so a handle does not wait for another
 * handle's sweep.  (Measured, round 5: an MI355X runs two such launch chains at full rate; a third waits.)
 *
 * Validate the problem, copy tables and knots to `device`, pick a kernel. */
int32_t hjb_create(const hjb_problem *problem, int32_t device, hjb_handle *out);
int32_t hjb_destroy(hjb_handle h);
/* text of the last error on this handle (h may be NULL: last create error) */
const char *hjb_last_error(hjb_handle h);
int32_t hjb_get_info(hjb_handle h, hjb_info *info);
/* tuning/testing knobs: "variant" (-1 automatic, 0..7 force a stage kernel; HJB_E_UNSUPPORTED when it does not
 * apply), "graph" (0/1: hipGraph replay inside hjb_solve), "temporal" (several stages per launch inside hjb_solve
 * for local 2-D problems, kernels_tile2d.h: 0 off, 1 when applicable [default], 2 required), "row_lean" (0/1: lean form of
 * stage kernel 6), "lds_pad" (extra dynamic LDS bytes per workgroup: occupancy experiments), "monitor_single" (0/1:
 * hjb_solve_opts.monitor_single for callers of hjb_solve_flat); read-only: "idx_bytes" */
int32_t hjb_set_option(hjb_handle h, const char *key, int64_t value);
/* "prep_mfma" (set): rebuild the handle's stage-invariant (cell, weight) tables - 1: with v_mfma_f32_32x32x2_f32 where an
 * axis' next-state sum splits into (all terms but the last) + (last term) over disjoint grid dims (the affine A x + B u
 * of every reference solver; csrc/kernels_prep_mfma.h), 0: with the vector kernels.  Bit-identical tables either way;
 * get: "prep_mfma", "prep_mfma_tables", "prep_tables", "prep_ns" (device time of the last rebuild), "table_hash",
 * "packed2_mode" (variant 4's contraction mode, csrc/kernels_packed2.h: 0 plain, 1 / 4 the C2 shape with / without the axis-0
 * table, 2 / 3 the per-state window of the attitude shapes with four last-axis planes, 5 / 6 with three - chosen when the inner
 * control moves the last axis by less than its narrowest cell per step; -1 when variant 4 does not apply).  Settable:
 * "window_planes" (3 or 4: switch a handle that qualifies between modes 5 / 6 and 2 / 3; same results, tests and timing).
 * read a knob back, plus what the column-sweep kernel (variant 7) settled on: "cs_dpp" (1: one load per corner row, the
 * upper axis-0 neighbour taken from the next lane), "cs_groups", "cs_group_axis", "cs_rows" (corner rows per step of the
 * mid-grid column), "cs_coop" (1: the cooperative form is in effect), "cs_coop_why" (why it does not apply: 0 applies,
 * 1 groups, 2 axis 1 sees the window axis, 3 n0 / storage, 4 cells, 5 window knots, 6 axis-0 knots).  Settable (testing,
 * tuning): "cs_dpp", "cs_coop" (1: eight neighbouring columns share their corner rows through LDS, kernels_colcoop.h;
 * off by default - slower on C4), "cs_xcd_mod" (residue modulus of the column -> XCD assignment: 0/1 contiguous ranges,
 * -1 the group spacing), "cs_xcd_axis" (0: the XCDs split the group axis [default], 1: the window axis), "cs_split" (parts
 * a column is swept in - each by a wave of its own, priming where it starts; 0 = automatic: more parts for launches with
 * few columns, e.g. the boundary strips of a multi-GPU slab; reads back the value in effect).  Variant 5: "tabled_i32"
 * (1 [default where every index of the problem fits 31 bits]: the table kernel's 32-bit form, 0: its general 64-bit form;
 * reads back the form a launch would take).  "grid" (get: workgroups per launch; set, timing experiments on the grid-stride
 * kernels only: the library's own choice walks a grid larger than the launch in equally long spans - until the next option that
 * re-chooses the launch).  "block" (set, variant 3 only: 256 / 512 / 1024 threads per workgroup = 64 x the states it sweeps side by
 * side; 512 by default where J is staged in LDS). */
int32_t hjb_get_option(hjb_handle h, const char *key, int64_t *value);

/* ONE backup, host buffers (exactly the MATLAB statement above):
 * J_next [j_elems] -> J_out [j_elems] (owned planes written), idx_out [n_states]. */
int32_t hjb_backup_stage(hjb_handle h, const void *J_next, void *J_out, void *idx_out);
/* ONE backup on device buffers, asynchronous on `stream` (a hipStream_t; NULL =
 * default stream).  For host-driven loops and multi-GPU halo exchange. */
int32_t hjb_backup_stage_device(hjb_handle h, const void *dJ_next, void *dJ_out,
                                void *d_idx_out, void *stream);
/* Non-zero if a previous device-side backup hit HJB_E_HALO (synchronises). */
int32_t hjb_check_device_status(hjb_handle h, void *stream);

/* ---- device-buffer helpers --------------------------------------------------------------------------------------
 * hjb_backup_stage_device works on device buffers the CALLER owns.  A host without a HIP binding of its own (MATLAB
 * through calllib, plain C) gets them here - enough to drive a grid that never exists in host memory (C3: 51^6 states,
 * 70 GB per buffer).  Pointers returned by hjb_device_malloc are ordinary HIP device pointers (hipMalloc). */
#define HJB_COPY_H2D 0
#define HJB_COPY_D2H 1
#define HJB_COPY_D2D 2
int32_t hjb_device_malloc(int32_t device, int64_t bytes, void **out);
int32_t hjb_device_free(int32_t device, void *p);
int32_t hjb_device_mem_info(int32_t device, int64_t *free_bytes, int64_t *total_bytes);
int32_t hjb_device_copy(int32_t device, void *dst, const void *src, int64_t bytes, int32_t kind);   /* synchronous */
/* dJ[s] = ((v0[i0] + v1[i1]) + v2[i2]) + ...  over the handle's whole grid: one add of the arithmetic type per axis,
 * axis 0 first, stored in the handle's J storage type.  vecs[a]: HOST vector of n[a] elements of the arithmetic type
 * (float for HJB_F32 / HJB_F16S).  A separable terminal cost for grids too large to build on the host.  On a SLAB handle dJ is
 * the slab's haloed buffer and is filled with the planes [slab_begin - halo_lo, slab_end + halo_hi) of that global function. */
int32_t hjb_device_fill_separable(hjb_handle h, const void *const *vecs, void *dJ, void *stream);
/* out[i] = d_src[sel[i]] for elements of elem_bytes (1, 2, 4, 8) bytes: sample a device-resident J or label array */
int32_t hjb_device_gather(int32_t device, const void *d_src, int32_t elem_bytes, const int64_t *sel, int64_t n_sel, void *out);

/* The whole backward sweep (the `for k` loops of the reference). */
int32_t hjb_solve(hjb_handle h, const hjb_solve_opts *opts, hjb_result *result);
/* n independent problems of ONE kernel shape swept side by side with ONE launch per stage for all of them: the four channels of
 * Solver_pos_att.simplified_run (pos-att/Solver_pos_att.m:197-242: 2.7e5 states each - a stage kernel of one channel is a launch
 * boundary plus one wave's chain of round trips, and of four such chains on four streams the device runs two at full rate).
 * Every problem keeps its own terminal cost, outputs, monitor sums / difference / stop decision (a stopped problem drops out of
 * the launches that follow) and progress callback; its results equal hjb_solve's bit for bit.  Conditions (else
 * HJB_E_UNSUPPORTED, and the caller sweeps the problems with hjb_solve on threads of their own): n <= 8, one device, one n_stages
 * and one monitor_period for all, no per-stage outputs / probe / per-stage progress, and every handle on ONE of
 *   - the column-sweep kernel in its usual form (float32 J, state cost terms + one control term, the one-load form), one group
 *     axis and one cost typing; the columns are cut into parts for the whole batch unless option "cs_split" is set;
 *   - the table kernel's 32-bit form (kernel_variant 5, every index below 2^31), one dtype and one D <= 4.  Worth it while all the
 *     problems' states are resident at once (<= ~5e5 states in all: a launch-bound stage); larger ones overlap better as chains
 *     of their own (profiles/r06_batch_attitude.log). */
int32_t hjb_solve_batch(int32_t n, const hjb_handle *handles, const hjb_solve_opts *const *opts, hjb_result *const *results);

/* Batched evaluation of a gridded function at nq points - what the reference does with the
 * sweep's results: griddedInterpolant({grid vectors}, U_vector(U_idx), 'nearest') policy lookups
 * (Solver_position.m:144-146, Solver_pos_att.m:851-861, :404-449) and 'linear' lookups of
 * u_star(:,:,k) / J (Dynamic_Solver.m:132-135).  values: [n_1..n_D] column-major, dtype;
 * queries: [D x nq] column-major (point i = queries[D*i .. D*i+D-1]), dtype; out: [nq] dtype.
 * method HJB_LOOKUP_LINEAR: N-linear with linear extrapolation (same arithmetic as the sweep);
 * HJB_LOOKUP_NEAREST: per axis the nearer knot of the enclosing cell, the upper one at the midpoint. */
#define HJB_LOOKUP_NEAREST 0
#define HJB_LOOKUP_LINEAR 1
int32_t hjb_policy_lookup(int32_t device, int32_t dtype, int32_t D, const int32_t *n, const double *const *knots,
                          const void *values, int64_t nq, const void *queries, int32_t method, void *out);

/* One stage of the probe block from a host J_next (for host-driven stage loops): same outputs as hjb_probe, one
 * plane each (g [B], x_next [B, D], j_interp [B]).  J_next may be NULL when j_interp is NULL. */
int32_t hjb_probe_stage(hjb_handle h, const void *J_next, const hjb_probe *probe);

/* ---- flat builder API -------------------------------------------------------------------------------------------
 * hjb_problem holds arrays of structs with pointers, which MATLAB's loadlibrary/calllib cannot marshal.  These entry
 * points take primitives and plain arrays only, copy what they are given (the caller may free it at once), and end in
 * hjb_create_from, after which the handle is used exactly like one from hjb_create.  They replace, for a MATLAB host,
 * the table building of the reference's run methods (test/Dynamic_Solver.m:66-84: grid vectors, a_D_M, g_D):
 *   hjb_problem_new(D, C, n, m, dtype, index_base, &b)
 *   hjb_problem_set_knots(b, axis, knots, len)                      once per state axis
 *   hjb_problem_add_next_term(b, axis, mask, data, count)           in MATLAB's left-to-right order of the sum
 *   hjb_problem_add_cost_term(b, mask, data, count)
 *   [hjb_problem_set_types(b, idx_dtype, table_dtype)]              right after hjb_problem_new
 *   [hjb_problem_set_slab(b, begin, end, halo_lo, halo_hi)]  [hjb_problem_set_model(b, model, h, t0, t1, t2, t3)]
 *   hjb_create_from(b, device, &h);  hjb_problem_free(b)
 * data: `count` elements of the problem dtype (float for HJB_F32 / HJB_F16S, double for HJB_F64), column-major over
 * the masked grid dims; count must equal the product of their sizes. */
typedef struct hjb_builder_s *hjb_builder;
int32_t hjb_problem_new(int32_t D, int32_t C, const int32_t *n, const int32_t *m, int32_t dtype, int32_t index_base,
                        hjb_builder *out);
int32_t hjb_problem_set_knots(hjb_builder b, int32_t axis, const double *knots, int32_t len);
int32_t hjb_problem_add_next_term(hjb_builder b, int32_t axis, uint32_t mask, const void *data, int64_t count);
int32_t hjb_problem_add_cost_term(hjb_builder b, uint32_t mask, const void *data, int64_t count);
int32_t hjb_problem_set_slab(hjb_builder b, int32_t slab_begin, int32_t slab_end, int32_t halo_lo, int32_t halo_hi);
/* hjb_problem.idx_dtype (HJB_IDX_*) and hjb_problem.table_dtype (HJB_TAB_*).  Call it right after hjb_problem_new:
 * with HJB_TAB_F64 the `data` of every hjb_problem_add_next_term that follows is float64 (cost terms stay float). */
int32_t hjb_problem_set_types(hjb_builder b, int32_t idx_dtype, int32_t table_dtype);
/* hjb_problem.cost_dtype (HJB_COST_*): call it before the first hjb_problem_add_cost_term - under HJB_COST_F64 the cost
 * terms' data are float64 */
int32_t hjb_problem_set_cost_type(hjb_builder b, int32_t cost_dtype);
int32_t hjb_problem_set_model(hjb_builder b, int32_t model, double model_h, const void *t0, const void *t1,
                              const void *t2, const void *t3);
/* Relabel the state axes of a problem under construction: new axis i = old axis order[i] (terms over several state
 * dims are transposed).  Which axis is last decides which stage kernel applies and which axis a multi-GPU run shards;
 * the caller permutes its own arrays the same way (MATLAB: permute(J, order + 1) in, ipermute out).  Before
 * hjb_problem_set_slab; not for problems with a state model. */
int32_t hjb_problem_permute_axes(hjb_builder b, const int32_t *order);
/* A labelling under which a faster stage kernel applies, from the terms' masks (and, for D = 4 with one control dim, the
 * control terms' ranges): order_out[D], *found = 1 when it differs from the present labelling (else the identity, 0).
 * D = 4, C = 1: the column-sweep kernel's shape (Solver_pos_att.m:299-328: the two axes the thrusters do not drive
 * first, of the other two the less-moved one last; 120^4 x 9: 1.8 ms per stage against 6.7 ms in the reference's own
 * order (x, v, theta, w)).  Otherwise: the axes no control drives first, then the driven ones in the order of the
 * control dims (Solver_attitude.m's (w1, w2, w3, yaw, pitch, roll) -> (yaw, pitch, roll, w1, w2, w3): 4.4 ms against
 * 28 ms on the reference grid). */
int32_t hjb_problem_suggest_order(hjb_builder b, int32_t *order_out, int32_t *found);
int32_t hjb_create_from(hjb_builder b, int32_t device, hjb_handle *out);
int32_t hjb_problem_free(hjb_builder b);
/* text of the last error of a builder call (b may be NULL) */
const char *hjb_problem_last_error(hjb_builder b);

/* hjb_solve without structs: every argument a scalar or a plain array (NULL where hjb_solve_opts allows NULL);
 * the three result scalars may be NULL.  No progress callback, no probe. */
int32_t hjb_solve_flat(hjb_handle h, int32_t n_stages, int32_t monitor_period, double monitor_tol, const void *terminal,
                       void *J_final, void *idx_final, void *J_stages, void *idx_stages, int32_t *stages_done,
                       int32_t *stopped_early, double *sweep_ms);
/* hjb_get_info without the struct: out[0..7] = n_states, n_controls, j_elems, kernel_variant, lds_bytes, grid,
 * halo_needed_lo, halo_needed_hi */
int32_t hjb_get_info_flat(hjb_handle h, int64_t *out8);

/* ---- single-process multi-GPU sweep -------------------------------------------------------------------------------
 * For a host that is ONE process (MATLAB): the stage loop of the reference (pos-att/Solver_pos_att.m:270-286,
 * position-control/Solver_position.m:132-141) over a grid partitioned along its LAST state axis, slab i on device
 * devices[i].  Per stage each slab gets the halo planes of J_{k+1} from its neighbours (hipMemcpyPeerAsync, xGMI), runs
 * the interior planes while the copies are in flight and the two boundary strips after them; the early-stop monitor's
 * sums are added over the slabs.  Choose WHICH axis is last by relabelling the state axes (the axis should move less
 * than a slab per stage: pos-att shards x, v or theta, not w).  hjb_solve_opts: terminal / J_final / idx_final are
 * whole-grid host arrays, as are J_stages / idx_stages (copied out slab by slab behind each stage); probe is not
 * supported.  The same device may be listed more than
 * once (several slabs on one GPU: testing).  One process per GPU with torch.distributed/RCCL is the other supported
 * form (hjbdp/sharded.py, bench.py --gpus N). */
typedef struct hjb_multi_s *hjb_multi;
int32_t hjb_create_multi(const hjb_problem *problem, int32_t n_dev, const int32_t *devices, hjb_multi *out);
int32_t hjb_create_multi_from(hjb_builder b, int32_t n_dev, const int32_t *devices, hjb_multi *out);
int32_t hjb_solve_multi(hjb_multi m, const hjb_solve_opts *opts, hjb_result *result);
int32_t hjb_solve_multi_flat(hjb_multi m, int32_t n_stages, int32_t monitor_period, double monitor_tol, const void *terminal,
                             void *J_final, void *idx_final, int32_t *stages_done, int32_t *stopped_early, double *sweep_ms);
/* planes [begin, end) of slab `slab`, its halo, whether it runs as interior + strips, its stage kernel (any out may be NULL) */
int32_t hjb_multi_slab_info(hjb_multi m, int32_t slab, int32_t *begin, int32_t *end, int32_t *halo_lo, int32_t *halo_hi,
                            int32_t *split, int32_t *kernel_variant);
int32_t hjb_multi_set_option(hjb_multi m, const char *key, int64_t value);   /* hjb_set_option on every slab handle */
int32_t hjb_destroy_multi(hjb_multi m);
const char *hjb_multi_last_error(hjb_multi m);

/* ---- one process per GPU: a rank's share of the sweep ----------------------------------------------------------------
 * For hosts that run one process per GPU and move the halo planes themselves (MPI; RCCL through torch.distributed:
 * hjbdp/sharded.py, bench.py --gpus N).  hjb_rank_create partitions the LAST state axis over `world` ranks exactly as
 * hjb_create_multi does over devices and builds this rank's handles: the slab, and - with `overlap` and an interior -
 * the interior and the two boundary strips over the same buffers.  The caller owns two J buffers of
 * (end - begin + halo_lo + halo_hi) planes and a label buffer of (end - begin) planes on `device`, and per stage
 *   1. makes its transfer stream wait for the compute stream, enqueues the halo exchange of dJ_in on it
 *      (send my boundary planes, receive into planes [0, halo_lo) and [halo_lo + owned, ...)),
 *   2. calls hjb_rank_stage(r, dJ_in, dJ_out, d_idx, compute_stream, halo_stream): ONE call enqueues the interior on
 *      compute_stream, the strips on the library's own streams behind an event recorded on halo_stream at this point,
 *      and joins them into compute_stream (halo_stream NULL: nothing to wait for).
 * hjb_rank_info: out10 = begin, end, halo_lo, halo_hi, split (1: interior + strips), kernel variant, halo_needed_lo,
 * halo_needed_hi, bytes per label, planes of the last axis. */
typedef struct hjb_rank_s *hjb_rank;
int32_t hjb_rank_create(const hjb_problem *problem, int32_t device, int32_t rank, int32_t world, int32_t overlap, hjb_rank *out);
/* the same from a flat builder (include/hjbdp_matlab.h: a MATLAB worker per GPU binds this one) */
int32_t hjb_rank_create_from(hjb_builder b, int32_t device, int32_t rank, int32_t world, int32_t overlap, hjb_rank *out);
int32_t hjb_rank_info(hjb_rank r, int32_t *out10);
int32_t hjb_rank_stage(hjb_rank r, const void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream, void *halo_stream);
/* The stage with the boundary strips enqueued FIRST, behind what halo_stream holds (the previous exchange), and
 * hjb_rank_wait_strips: `stream` waits for that stage's strips (*covered = 1 when they cover every plane a neighbour needs;
 * 0: nothing is enqueued, the caller's exchange waits for the compute stream).  The pieces of hjb_rank_step_post (below) for a
 * host that moves the halo planes itself: exchange dJ_out's boundary planes behind the strips, under the rest of the interior. */
int32_t hjb_rank_stage_post(hjb_rank r, const void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream, void *halo_stream);
int32_t hjb_rank_wait_strips(hjb_rank r, void *stream, int32_t *covered);
int32_t hjb_rank_set_option(hjb_rank r, const char *key, int64_t value);
int32_t hjb_rank_get_option(hjb_rank r, const char *key, int64_t *value);
int32_t hjb_rank_check_status(hjb_rank r, void *stream);
/* dJ (this rank's haloed buffer) = the separable function of hjb_device_fill_separable on the rank's planes, halo planes included:
 * vecs[a] are the GLOBAL host vectors (n[a] elements each).  A terminal cost for grids that never exist in host memory. */
int32_t hjb_rank_fill_separable(hjb_rank r, const void *const *vecs, void *dJ, void *stream);
int32_t hjb_rank_destroy(hjb_rank r);
const char *hjb_rank_last_error(hjb_rank r);

/* ---- RCCL inside the library: a rank's halo exchange and monitor all-reduce without torch or MPI -------------------------
 * per stage ncclGroupStart; ncclSend / ncclRecv x <= 4; ncclGroupEnd on a transfer stream (one xGMI link
 * per neighbour pair), every monitor period a 2-double ncclAllReduce.  librccl.so.1 is dlopen'ed on first use (override:
 * $HJBDP_RCCL_LIB); libhjbdp has no link dependency on it.  The host's part: rank 0 calls hjb_rank_comm_unique_id and hands
 * the 128 bytes to every rank by any means (a file, a socket, MATLAB's labSend); every rank calls hjb_rank_comm_init
 * (collective), then either hjb_rank_sweep (the whole `for k_s` loop incl. the monitor of Solver_pos_att.m:268-285) or,
 * stage by stage, hjb_rank_step = hjb_rank_exchange (transfer stream, behind the compute stream) + hjb_rank_stage with the
 * boundary strips waiting for the halos.  hjb_rank_monitor_sums: sum J and sum of labels over the WHOLE grid on every rank.
 * Option "comm_loopback" (hjb_rank_set_option before hjb_rank_comm_init; one-GPU transport test): a communicator of one
 * rank whose two neighbours are itself - the planes it sends down arrive in its own upper halo, those it sends up in its
 * lower halo - so the RCCL calls, pointers, counts and stream ordering run on a box with a single GPU.  Option
 * "xfer_delay_us" (emulation only, tools/emulate_ranks.py): a spin of that many microseconds on the transfer stream behind
 * every exchange, standing in for link latency the one-GPU loopback does not have.
 * Option "monitor_single" (the reference's typing, Solver_pos_att.m:276-282): hjb_rank_sweep forms `e = fsum50 - fsum50_prev`
 * and `abs(e) < tol` in single precision, as hjb_solve does.  The SUM itself: at world == 1 the library's stated float32
 * tree (= hjb_solve's, bit for bit); over several ranks each rank's planes are summed in float64 over a fixed tree and
 * the all-reduce adds the ranks in its own order - reproducible for one world size, NOT bit-identical to the one-device
 * sum, so a stop decision within rounding of `tol` can fall one monitor period apart.
 * A host without librccl: hjb_rank_comm_unique_id / hjb_rank_comm_init return HJB_E_UNSUPPORTED with the loader's message.
 * tools/bench_ranks.cpp is a C++ driver on these calls (one process per GPU, no Python).
 * hjb_rank_comm_available: HJB_OK when the library can reach RCCL (loads it on first use, nothing else - no communicator, no
 * bootstrap thread), else HJB_E_UNSUPPORTED with the loader's message in hjb_rank_last_error(NULL): what EVERY rank asks before
 * the collective hjb_rank_comm_init, so that the ranks can agree to use another transport instead of one of them failing alone. */
int32_t hjb_rank_comm_available(void);
int32_t hjb_rank_comm_unique_id(void *id128_out);
int32_t hjb_rank_comm_init(hjb_rank r, const void *id128);
/* what the communicator itself reports (ncclCommCount / ncclCommUserRank; -1 where librccl lacks the query): lets a run
 * verify that the ranks it timed were ranks of ONE communicator of the expected size */
int32_t hjb_rank_comm_info(hjb_rank r, int32_t *n_ranks, int32_t *comm_rank);
int32_t hjb_rank_exchange(hjb_rank r, void *dJ, void *compute_stream);
void *hjb_rank_transfer_stream(hjb_rank r);
int32_t hjb_rank_step(hjb_rank r, void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream);
/* The stage in the order that hides the exchange: boundary strips FIRST (the halos of dJ_in arrived during the previous stage),
 * the interior beside them, and the exchange of dJ_OUT's boundary planes as soon as the strips are done, under the rest of the
 * interior.  Precondition: dJ_in's halos are valid - one hjb_rank_exchange(r, dJ_in, stream) before the first step; every step
 * leaves dJ_out's halos filled (ordered for the next hjb_rank_step_post / the transfer stream / a device synchronisation).
 * hjb_rank_sweep runs this form (option "post_exchange" 0: hjb_rank_step) and returns - and stops its clock - only behind the
 * last stage's exchange, so the final buffer's halo planes are at rest when it hands the buffer back.
 * CONCURRENT USE OF THE COMMUNICATOR: a rank has ONE communicator; the exchange runs on the transfer stream, the monitor's
 * all-reduce on the compute stream.  The library never has two RCCL operations of one rank in flight unordered: the monitor's
 * all-reduce is enqueued behind an event that follows the pending exchange (hjb_rank_monitor_sums waits for it on the compute
 * stream), and the next exchange is ordered behind the compute stream.  A host that issues its own RCCL calls on this
 * communicator must keep the same rule - one stream order per communicator. */
int32_t hjb_rank_step_post(hjb_rank r, void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream);
int32_t hjb_rank_monitor_sums(hjb_rank r, const void *dJ, const void *d_idx, void *compute_stream, double *sums2);
int32_t hjb_rank_sweep(hjb_rank r, int32_t n_stages, int32_t monitor_period, double monitor_tol, void *dJ0, void *dJ1, void *d_idx,
                       void *compute_stream, int32_t *stages_done, int32_t *stopped_early, int32_t *final_in_0, double *sweep_ms);

#ifdef __cplusplus
}
#endif
#endif /* HJBDP_H */
