/* hjbdp_matlab.h - the flat part of include/hjbdp.h for MATLAB's loadlibrary.
 *
 * loadlibrary parses C prototypes but cannot marshal hjb_problem (arrays of structs holding pointers) or incomplete
 * struct pointer types comfortably.  Every function below takes primitives, plain arrays and opaque `void *` handles
 * only; the symbols are the same ones include/hjbdp.h declares (hjb_builder / hjb_handle / hjb_multi are pointers,
 * spelled void * here).  Usage: matlab/hjbdp_solve.m; semantics: include/hjbdp.h.
 *
 *   loadlibrary('libhjbdp.so', 'hjbdp_matlab.h')
 */
#ifndef HJBDP_MATLAB_H
#define HJBDP_MATLAB_H
#include <stdint.h>

const char *hjb_version(void);
const char *hjb_status_string(int32_t status);
int32_t hjb_device_count(void);

/* problem description: replaces the table building of the reference's run methods (test/Dynamic_Solver.m:66-84) */
int32_t hjb_problem_new(int32_t D, int32_t C, const int32_t *n, const int32_t *m, int32_t dtype, int32_t index_base, void **builder_out);
int32_t hjb_problem_set_knots(void *builder, int32_t axis, const double *knots, int32_t len);
int32_t hjb_problem_add_next_term(void *builder, int32_t axis, uint32_t mask, const void *data, int64_t count);
int32_t hjb_problem_add_cost_term(void *builder, uint32_t mask, const void *data, int64_t count);
int32_t hjb_problem_set_slab(void *builder, int32_t slab_begin, int32_t slab_end, int32_t halo_lo, int32_t halo_hi);
int32_t hjb_problem_set_types(void *builder, int32_t idx_dtype, int32_t table_dtype);
int32_t hjb_problem_set_cost_type(void *builder, int32_t cost_dtype);
int32_t hjb_problem_set_model(void *builder, int32_t model, double model_h, const void *t0, const void *t1, const void *t2, const void *t3);
int32_t hjb_problem_permute_axes(void *builder, const int32_t *order);
int32_t hjb_problem_suggest_order(void *builder, int32_t *order_out, int32_t *found);
int32_t hjb_create_from(void *builder, int32_t device, void **handle_out);
int32_t hjb_problem_free(void *builder);
const char *hjb_problem_last_error(void *builder);

/* the stage loops: test/Dynamic_Solver.m:86-102, Solver_position.m:132-141, Solver_attitude.m:236-247 / :280-287,
 * Solver_pos_att.m:270-286 */
int32_t hjb_solve_flat(void *handle, int32_t n_stages, int32_t monitor_period, double monitor_tol, const void *terminal,
                       void *J_final, void *idx_final, void *J_stages, void *idx_stages, int32_t *stages_done,
                       int32_t *stopped_early, double *sweep_ms);
/* one stage: [F.Values, idx] = min(J_stage + F(x_next...), [], ctrl_dim)  (Dynamic_Solver.m:207-210) */
int32_t hjb_backup_stage(void *handle, const void *J_next, void *J_out, void *idx_out);
/* ... on device buffers (hjb_device_malloc below), asynchronous: the entry point of a host that keeps its own `for k` loop
 * (hjbdp_solve.m 'on_stage'); stream NULL = the default stream.  hjb_check_device_status synchronises and reports a left slab. */
int32_t hjb_backup_stage_device(void *handle, const void *dJ_next, void *dJ_out, void *d_idx_out, void *stream);
int32_t hjb_check_device_status(void *handle, void *stream);
int32_t hjb_get_info_flat(void *handle, int64_t *out8);
int32_t hjb_set_option(void *handle, const char *key, int64_t value);
int32_t hjb_get_option(void *handle, const char *key, int64_t *value);
const char *hjb_last_error(void *handle);
int32_t hjb_destroy(void *handle);

/* the same loop over several GPUs of this process (slabs of the last state axis) */
int32_t hjb_create_multi_from(void *builder, int32_t n_dev, const int32_t *devices, void **multi_out);
int32_t hjb_solve_multi_flat(void *multi, int32_t n_stages, int32_t monitor_period, double monitor_tol, const void *terminal,
                             void *J_final, void *idx_final, int32_t *stages_done, int32_t *stopped_early, double *sweep_ms);
int32_t hjb_multi_set_option(void *multi, const char *key, int64_t value);
const char *hjb_multi_last_error(void *multi);
int32_t hjb_destroy_multi(void *multi);

/* one process (MATLAB worker) per GPU: this rank's slab of the last state axis, one call per stage; the worker owns the
 * device buffers (hjb_device_*) and moves the halo planes between stages (include/hjbdp.h "one process per GPU") */
int32_t hjb_rank_create_from(void *builder, int32_t device, int32_t rank, int32_t world, int32_t overlap, void **rank_out);
int32_t hjb_rank_info(void *rank, int32_t *out10);
int32_t hjb_rank_stage(void *rank, const void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream, void *halo_stream);
int32_t hjb_rank_set_option(void *rank, const char *key, int64_t value);
int32_t hjb_rank_get_option(void *rank, const char *key, int64_t *value);
int32_t hjb_rank_check_status(void *rank, void *stream);
int32_t hjb_rank_destroy(void *rank);
const char *hjb_rank_last_error(void *rank);
/* RCCL transport inside the library (include/hjbdp.h): the 128-byte id from rank 0 goes to every worker by labSend / a file */
int32_t hjb_rank_comm_available(void);
int32_t hjb_rank_comm_unique_id(void *id128_out);
int32_t hjb_rank_comm_init(void *rank, const void *id128);
int32_t hjb_rank_step(void *rank, void *dJ_in, void *dJ_out, void *d_idx, void *compute_stream);
int32_t hjb_rank_monitor_sums(void *rank, const void *dJ, const void *d_idx, void *compute_stream, double *sums2);
int32_t hjb_rank_sweep(void *rank, int32_t n_stages, int32_t monitor_period, double monitor_tol, void *dJ0, void *dJ1, void *d_idx,
                       void *compute_stream, int32_t *stages_done, int32_t *stopped_early, int32_t *final_in_0, double *sweep_ms);
/* device buffers for hosts without a HIP binding of their own */
int32_t hjb_device_malloc(int32_t device, int64_t bytes, void **out);
int32_t hjb_device_free(int32_t device, void *p);
int32_t hjb_device_copy(int32_t device, void *dst, const void *src, int64_t bytes, int32_t kind);

/* griddedInterpolant(..., 'nearest' | 'linear') lookups of the results (Solver_position.m:144-146, Dynamic_Solver.m:132-135) */
int32_t hjb_policy_lookup(int32_t device, int32_t dtype, int32_t D, const int32_t *n, const double *const *knots,
                          const void *values, int64_t nq, const void *queries, int32_t method, void *out);
#endif
