/* hjbdp_mex.c - MEX gateway to libhjbdp (include/hjbdp.h), for MATLAB installations where loadlibrary is not an
 * option.  One entry point, the call sequence of matlab/hjbdp_solve.m:
 *
 *     out = hjbdp_mex(prob, n_stages, opts)
 *
 *   prob.knots       1xD cell of grid vectors (double)                       test/Dynamic_Solver.m:69
 *   prob.m           1xC control grid sizes
 *   prob.single      logical: single (Dynamic_Solver.m:69) or double (test_coder.m) arithmetic
 *   prob.next_terms  1xD cell; next_terms{a} is a struct array with fields dims (1-based grid dims the operand varies
 *                    along: states 1..D, controls D+1..D+C) and data (the operand, any shape, column-major) - the
 *                    summands of x_next_a in MATLAB's left-to-right order (a_D_M, Dynamic_Solver.m:184-188)
 *   prob.cost_terms  struct array (dims, data): the summands of the stage cost (g_D, :196-200)
 *   prob.terminal    optional terminal cost [nS] (default zeros, :83-84)
 *   opts             struct, all fields optional: keep_stages (false), monitor_period (0), monitor_tol (0), devices (0;
 *                    a vector partitions the last state axis over those GPUs: hjb_create_multi_from), double_tables
 *                    (false; true with prob.single: next_terms data stay double and the queries are located and weighted
 *                    in double - the typing of pos-att/Solver_pos_att.m:299-327, hjbdp.h HJB_TAB_F64), monitor_single
 *                    (false; true: the monitor's sum(F.Values(:)) is a single-precision sum, Solver_pos_att.m:274).
 *                    The state axes are never relabelled here (hjbdp_solve.m's 'fast_axes' is off by default too).
 *   out              struct: J, idx (double, 1-based, as MATLAB's min returns), J_stages, idx_stages ([nS x n_stages],
 *                    stage k_s in column k_s, only with keep_stages), stages_done, stopped_early, sweep_ms
 *
 * Replaces the stage loops test/Dynamic_Solver.m:86-102, position-control/Solver_position.m:132-141,
 * attitude-control/Solver_attitude.m:236-247 / :280-287, pos-att/Solver_pos_att.m:270-286.
 *
 * Build where MATLAB exists:  mex -I<repo>/include mex/hjbdp_mex.c -L<repo>/optimal-control-dynamic-programming_amd/hjbdp -lhjbdp
 * The build image has no MATLAB (no mex.h): this file is syntax-checked against a minimal declaration stub
 * (tests/mex_stub/mex.h, test-only) and is otherwise unexecuted; the same call sequence is executed through ctypes
 * by tests/test_gpu_flat_api.py.
 */
#include <stdint.h>
#include <string.h>

#include "mex.h"

#include "hjbdp.h"

static hjb_builder g_builder = NULL;
static hjb_handle g_handle = NULL;
static hjb_multi g_multi = NULL;

static void release_all(void) {
    if (g_handle) { hjb_destroy(g_handle); g_handle = NULL; }
    if (g_multi) { hjb_destroy_multi(g_multi); g_multi = NULL; }
    if (g_builder) { hjb_problem_free(g_builder); g_builder = NULL; }
}

static void fail(const char *id, const char *what, const char *detail) {
    /* mexErrMsgIdAndTxt does not return: free the native objects first */
    char buf[768];
    strncpy(buf, detail ? detail : "", sizeof buf - 1);
    buf[sizeof buf - 1] = 0;
    release_all();
    mexErrMsgIdAndTxt(id, "%s: %s", what, buf);
}

static const mxArray *need_field(const mxArray *s, const char *name) {
    const mxArray *f = mxIsStruct(s) ? mxGetField(s, 0, name) : NULL;
    if (!f) fail("hjbdp:arg", "missing field", name);
    return f;
}

static double opt_scalar(const mxArray *opts, const char *name, double dflt) {
    const mxArray *f = (opts && mxIsStruct(opts)) ? mxGetField(opts, 0, name) : NULL;
    return (f && mxGetNumberOfElements(f) >= 1) ? mxGetScalar(f) : dflt;
}

/* data of one term in the problem's arithmetic type; returns a buffer owned by MATLAB's allocator */
static void *term_data(const mxArray *data, int use_single, int64_t *count) {
    const size_t n = mxGetNumberOfElements(data);
    size_t i;
    *count = (int64_t)n;
    if (use_single) {
        float *v = (float *)mxMalloc(n * sizeof(float) + 4);
        if (mxIsSingle(data)) memcpy(v, mxGetData(data), n * sizeof(float));
        else if (mxIsDouble(data)) { const double *d = mxGetPr(data); for (i = 0; i < n; ++i) v[i] = (float)d[i]; }
        else fail("hjbdp:arg", "term data", "must be single or double");
        return v;
    } else {
        double *v = (double *)mxMalloc(n * sizeof(double) + 8);
        if (mxIsDouble(data)) memcpy(v, mxGetPr(data), n * sizeof(double));
        else if (mxIsSingle(data)) { const float *d = (const float *)mxGetData(data); for (i = 0; i < n; ++i) v[i] = d[i]; }
        else fail("hjbdp:arg", "term data", "must be single or double");
        return v;
    }
}

static uint32_t dims_to_mask(const mxArray *dims, int n_grid_dims) {
    const size_t n = mxGetNumberOfElements(dims);
    const double *d;
    uint32_t m = 0;
    size_t i;
    if (!mxIsDouble(dims)) fail("hjbdp:arg", "term dims", "must be double");
    d = mxGetPr(dims);
    for (i = 0; i < n; ++i) {                                  /* MATLAB dims are 1-based: 1 .. D + C */
        if (!(d[i] >= 1 && d[i] <= n_grid_dims) || d[i] != (double)(int)d[i]) fail("hjbdp:arg", "term dims", "must be integers in 1..D+C");
        m |= 1u << ((int)d[i] - 1);
    }
    return m;
}

static void add_terms(const mxArray *terms, int axis /* -1: the stage cost */, int use_single, int n_grid_dims) {
    const size_t nt = mxGetNumberOfElements(terms);
    size_t k;
    if (!mxIsStruct(terms)) fail("hjbdp:arg", "terms", "must be a struct array with fields dims, data");
    for (k = 0; k < nt; ++k) {
        const mxArray *dims = mxGetField(terms, k, "dims"), *data = mxGetField(terms, k, "data");
        int64_t count;
        void *v;
        int st;
        if (!dims || !data) fail("hjbdp:arg", "terms", "need fields dims and data");
        v = term_data(data, use_single, &count);
        st = axis < 0 ? hjb_problem_add_cost_term(g_builder, dims_to_mask(dims, n_grid_dims), v, count)
                      : hjb_problem_add_next_term(g_builder, axis, dims_to_mask(dims, n_grid_dims), v, count);
        mxFree(v);                                            /* the builder copied it */
        if (st) fail("hjbdp:problem", hjb_status_string(st), hjb_problem_last_error(g_builder));
    }
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    const mxArray *prob, *opts, *knots, *mfield, *nterms, *terminal;
    int32_t n[HJB_MAX_D] = {0}, m[HJB_MAX_C] = {0}, devices[64];
    int D, C, a, c, use_single, n_stages, keep_stages, monitor_period, n_dev = 1, st, double_tables, monitor_single;
    double monitor_tol;
    size_t nS = 1, esz;
    mxClassID cls;
    mxArray *J, *I, *Js = NULL, *Is = NULL;
    int32_t *idx32, *idxs32 = NULL, done = 0, early = 0;
    double ms = 0.0;
    void *term = NULL;
    const char *fields[] = {"J", "idx", "J_stages", "idx_stages", "stages_done", "stopped_early", "sweep_ms"};
    mwSize dims_out[HJB_MAX_D + 1];

    (void)nlhs;
    if (nrhs < 2) mexErrMsgIdAndTxt("hjbdp:arg", "usage: out = hjbdp_mex(prob, n_stages [, opts])");
    prob = prhs[0];
    opts = nrhs > 2 ? prhs[2] : NULL;
    if (mxGetNumberOfElements(prhs[1]) != 1 || !(mxGetScalar(prhs[1]) >= 1) || mxGetScalar(prhs[1]) > 2147483647.0)
        mexErrMsgIdAndTxt("hjbdp:arg", "n_stages must be a scalar >= 1");       /* before it sizes any allocation */
    n_stages = (int)mxGetScalar(prhs[1]);
    knots = need_field(prob, "knots");
    mfield = need_field(prob, "m");
    nterms = need_field(prob, "next_terms");
    if (!mxIsCell(knots) || !mxIsCell(nterms)) fail("hjbdp:arg", "prob", "knots and next_terms must be cell arrays");
    D = (int)mxGetNumberOfElements(knots);
    C = (int)mxGetNumberOfElements(mfield);
    if (D < 1 || D > HJB_MAX_D || C < 1 || C > HJB_MAX_C) fail("hjbdp:arg", "prob", "1..6 state axes, 1..3 control axes");
    if ((int)mxGetNumberOfElements(nterms) != D) fail("hjbdp:arg", "prob", "next_terms needs one entry per state axis");
    use_single = mxIsLogicalScalarTrue(need_field(prob, "single")) ? 1 : 0;
    for (a = 0; a < D; ++a) { n[a] = (int32_t)mxGetNumberOfElements(mxGetCell(knots, a)); nS *= (size_t)n[a]; }
    if (!mxIsDouble(mfield)) fail("hjbdp:arg", "prob.m", "must be a double vector");       /* mxGetPr is for double arrays only */
    for (c = 0; c < C; ++c) m[c] = (int32_t)mxGetPr(mfield)[c];
    keep_stages = opt_scalar(opts, "keep_stages", 0) != 0;
    monitor_period = (int)opt_scalar(opts, "monitor_period", 0);
    monitor_tol = opt_scalar(opts, "monitor_tol", 0);
    double_tables = opt_scalar(opts, "double_tables", 0) != 0;
    monitor_single = opt_scalar(opts, "monitor_single", 0) != 0;
    if (double_tables && !use_single) fail("hjbdp:arg", "opts.double_tables", "is for prob.single = true");
    devices[0] = 0;
    if (opts && mxIsStruct(opts) && mxGetField(opts, 0, "devices")) {
        const mxArray *dv = mxGetField(opts, 0, "devices");
        n_dev = (int)mxGetNumberOfElements(dv);
        if (n_dev < 1 || n_dev > 64 || !mxIsDouble(dv)) fail("hjbdp:arg", "opts.devices", "1..64 devices (double)");
        for (a = 0; a < n_dev; ++a) devices[a] = (int32_t)mxGetPr(dv)[a];
    }
    if (n_dev > 1 && keep_stages) fail("hjbdp:arg", "opts", "keep_stages needs a single device");

    release_all();                                            /* leftovers of an interrupted call */
    st = hjb_problem_new(D, C, n, m, use_single ? HJB_F32 : HJB_F64, 1 /* MATLAB's 1-based argmin labels */, &g_builder);
    if (st) fail("hjbdp:problem", hjb_status_string(st), hjb_problem_last_error(NULL));
    if (double_tables) {
        st = hjb_problem_set_types(g_builder, HJB_IDX_I32, HJB_TAB_F64);
        if (st) fail("hjbdp:problem", hjb_status_string(st), hjb_problem_last_error(g_builder));
    }
    for (a = 0; a < D; ++a) {
        const mxArray *k = mxGetCell(knots, a);
        if (!mxIsDouble(k)) fail("hjbdp:arg", "knots", "must be double vectors");
        st = hjb_problem_set_knots(g_builder, a, mxGetPr(k), n[a]);
        if (st) fail("hjbdp:problem", hjb_status_string(st), hjb_problem_last_error(g_builder));
        add_terms(mxGetCell(nterms, a), a, use_single && !double_tables, D + C);
    }
    add_terms(need_field(prob, "cost_terms"), -1, use_single, D + C);

    cls = use_single ? mxSINGLE_CLASS : mxDOUBLE_CLASS;
    esz = use_single ? sizeof(float) : sizeof(double);
    terminal = mxIsStruct(prob) ? mxGetField(prob, 0, "terminal") : NULL;
    if (terminal && mxGetNumberOfElements(terminal) > 0) {
        int64_t cnt;
        if (mxGetNumberOfElements(terminal) != nS) fail("hjbdp:arg", "prob.terminal", "needs one value per state");
        term = term_data(terminal, use_single, &cnt);
    }
    for (a = 0; a < D; ++a) dims_out[a] = (mwSize)n[a];
    if (D == 1) dims_out[1] = 1;
    J = mxCreateNumericArray(D == 1 ? 2 : D, dims_out, cls, mxREAL);
    idx32 = (int32_t *)mxMalloc(nS * sizeof(int32_t) + 4);
    if (keep_stages) {
        Js = mxCreateNumericMatrix(nS, (mwSize)n_stages, cls, mxREAL);
        idxs32 = (int32_t *)mxMalloc(nS * (size_t)n_stages * sizeof(int32_t) + 4);
    }
    (void)esz;
    if (n_dev == 1) {
        st = hjb_create_from(g_builder, devices[0], &g_handle);
        if (st) fail("hjbdp:create", hjb_status_string(st), hjb_problem_last_error(g_builder));
        if (monitor_single) (void)hjb_set_option(g_handle, "monitor_single", 1);
        st = hjb_solve_flat(g_handle, n_stages, monitor_period, monitor_tol, term, mxGetData(J), idx32,
                            Js ? mxGetData(Js) : NULL, idxs32, &done, &early, &ms);
        if (st) fail("hjbdp:solve", hjb_status_string(st), hjb_last_error(g_handle));
    } else {
        st = hjb_create_multi_from(g_builder, n_dev, devices, &g_multi);
        if (st) fail("hjbdp:create", hjb_status_string(st), hjb_problem_last_error(g_builder));
        if (monitor_single) fail("hjbdp:arg", "opts.monitor_single", "needs a single device");
        st = hjb_solve_multi_flat(g_multi, n_stages, monitor_period, monitor_tol, term, mxGetData(J), idx32, &done, &early, &ms);
        if (st) fail("hjbdp:solve", hjb_status_string(st), hjb_multi_last_error(g_multi));
    }
    release_all();
    if (term) mxFree(term);

    /* MATLAB's min returns double indices (Dynamic_Solver.m:209-210) */
    I = mxCreateNumericArray(D == 1 ? 2 : D, dims_out, mxDOUBLE_CLASS, mxREAL);
    {
        double *d = mxGetPr(I);
        size_t i;
        for (i = 0; i < nS; ++i) d[i] = (double)idx32[i];
    }
    mxFree(idx32);
    if (keep_stages) {
        double *d;
        size_t i, tot = nS * (size_t)n_stages;
        Is = mxCreateNumericMatrix(nS, (mwSize)n_stages, mxDOUBLE_CLASS, mxREAL);
        d = mxGetPr(Is);
        for (i = 0; i < tot; ++i) d[i] = (double)idxs32[i];
        mxFree(idxs32);
    }
    plhs[0] = mxCreateStructMatrix(1, 1, 7, fields);
    mxSetField(plhs[0], 0, "J", J);
    mxSetField(plhs[0], 0, "idx", I);
    if (Js) mxSetField(plhs[0], 0, "J_stages", Js);
    if (Is) mxSetField(plhs[0], 0, "idx_stages", Is);
    mxSetField(plhs[0], 0, "stages_done", mxCreateDoubleScalar((double)done));
    mxSetField(plhs[0], 0, "stopped_early", mxCreateLogicalScalar(early != 0));
    mxSetField(plhs[0], 0, "sweep_ms", mxCreateDoubleScalar(ms));
}
