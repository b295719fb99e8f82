/*
 * prost_c.h -- command-level C ABI of the prost host library (libprost.so).
 *
 * Mirror of the reference's MEX gateway (matlab/+prost/private/prost.cpp:305-347): one entry
 * point, prost_command(cmd, nlhs, plhs, nrhs, prhs), with the same command table
 *     init, release, solve_problem, eval_linop, eval_prox, list_gpus, set_gpu
 * and the same argument layout (nested cells / structs / matrices).  MATLAB's mxArray is replaced
 * by the self-contained prost_value tree below, so that any host language (MATLAB MEX, Python
 * ctypes, cgo, JNI) can marshal its problem description into it -- see INTEGRATION.md for the
 * mxArray -> prost_value converter a MEX maintainer would add.
 *
 * Extra commands (no reference counterpart, used by the benchmark and the tests):
 *     set_precision / get_precision   'double' (reference default, config.hpp:7) or 'single'
 *     problem_info                    host-side problem setup only (no GPU needed)
 *     solver_create / solver_iterate / solver_state / solver_destroy
 *                                     persistent solver handle for timing K iterations
 *     comm_unique_id / comm_init / comm_destroy
 *                                     RCCL communicator for the global stopping criterion
 *     set_quirks                      reference bug-compatibility switches
 *
 * All functions are thread-compatible but, like the reference gateway, not re-entrant.
 */
#ifndef PROST_C_H_
#define PROST_C_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct prost_value prost_value;

enum {
  PROST_VALUE_EMPTY = 0,
  PROST_VALUE_MATRIX = 1,    /* dense real double matrix, column-major (scalars are 1x1)   */
  PROST_VALUE_STRING = 2,
  PROST_VALUE_CELL = 3,      /* 1-D list of values                                         */
  PROST_VALUE_STRUCT = 4,    /* named fields                                               */
  PROST_VALUE_SPARSE = 5,    /* real double sparse matrix in CSC (MATLAB Ir / Jc layout)   */
  PROST_VALUE_CALLBACK = 6   /* function handle (opts.interm_cb)                           */
};

/* interm callback: (user, iteration, x, nx, y, ny) -> non-zero = converged
 * (factory.cpp:136-158 calls feval(handle, it, x, y)) */
typedef int (*prost_interm_cb)(void* user, int iteration, const double* x, size_t nx, const double* y, size_t ny);
/* stop callback polled every iteration (prost.cpp:58-66, utIsInterruptPending) */
typedef int (*prost_stop_cb)(void* user);

/* ---- constructors (the tree owns its children; free the root with prost_value_free) ---- */
prost_value* prost_value_scalar(double v);
prost_value* prost_value_matrix(const double* data, size_t rows, size_t cols);      /* copies */
prost_value* prost_value_string(const char* s);
prost_value* prost_value_cell(size_t n);
int prost_value_cell_set(prost_value* cell, size_t i, prost_value* v);               /* takes ownership of v */
prost_value* prost_value_struct(void);
int prost_value_struct_set(prost_value* s, const char* name, prost_value* v);        /* takes ownership of v */
prost_value* prost_value_sparse(size_t rows, size_t cols, size_t nnz, const double* val, const int64_t* ir, const int64_t* jc);
prost_value* prost_value_callback(prost_interm_cb fn, void* user);
void prost_value_free(prost_value* v);

/* ---- accessors (borrowed pointers, valid while the value lives) ---- */
int prost_value_kind(const prost_value* v);
size_t prost_value_rows(const prost_value* v);
size_t prost_value_cols(const prost_value* v);
const double* prost_value_data(const prost_value* v);
const char* prost_value_str(const prost_value* v);
size_t prost_value_count(const prost_value* v);                                      /* cells */
const prost_value* prost_value_cell_get(const prost_value* v, size_t i);
const prost_value* prost_value_field(const prost_value* v, const char* name);        /* NULL if absent */
size_t prost_value_field_count(const prost_value* v);                                /* struct fields, in insertion order  */
const char* prost_value_field_name(const prost_value* v, size_t i);                  /* NULL if i >= field_count           */

/* ---- the gateway ---- */
/* Returns 0 on success.  plhs[0..nlhs) receive newly created values the caller must free.
 * On failure returns non-zero and prost_last_error() holds the message the reference would have
 * passed to mexErrMsgTxt (prost.cpp:342-346). */
int prost_command(const char* cmd, int nlhs, prost_value** plhs, int nrhs, const prost_value* const* prhs);
const char* prost_last_error(void);
/* Text the library prints (std::cout: the verbose header and the "It k: Feas_p=..." lines of solve_problem, list_gpus, the
 * |K| rescale note) goes to `fn(user, text, n)` -- n bytes, not NUL-terminated -- instead of the process's stdout: the
 * mexstream / scoped_redirect_cout of the MEX gateway (prost.cpp:15-44), which hands it to mexPrintf.  fn = NULL restores
 * stdout. */
typedef void (*prost_output_cb)(void* user, const char* text, size_t n);
void prost_set_output_callback(prost_output_cb fn, void* user);
/* user-interrupt hook of solve_problem (the Ctrl-C poll of the MEX gateway), asked once per kernel launch -- after every
 * iteration or every second one where two iterations share a launch, so a stop request is honoured at most one iteration
 * later than by the reference (solver.cu:151 polls after each).  fn = NULL removes it. */
void prost_set_stop_callback(prost_stop_cb fn, void* user);
/* Multi-rank runs without RCCL (several ranks on one GPU, or a host-side fabric): makes `fn(user, values, count)` -- an
 * in-place sum over the ranks of `count` doubles in pinned host memory -- the communicator of the solvers created
 * afterwards, exactly as comm_init does with an RCCL communicator (prost_hip_comm_create_host).  Returns 0 / 1
 * (prost_last_error); undone by the comm_destroy command. */
typedef void (*prost_allreduce_cb)(void* user, double* values, size_t count);
int prost_comm_init_host(prost_allreduce_cb fn, void* user, int world_size);
/* Point-to-point function of that communicator (halo columns of column-sharded images between the ranks:
 * solver_halo_exchange / solver_iterate_sharded): `fn(user, nops, is_send, peers, bufs, bytes)` performs all the transfers
 * of one exchange on pinned host buffers and returns when they are complete (prost_hip_comm_host_configure).  Call after
 * prost_comm_init_host. */
typedef void (*prost_p2p_cb)(void* user, int nops, const int* is_send, const int* peers, void* const* bufs, const size_t* bytes);
int prost_comm_set_host_p2p(prost_p2p_cb fn, void* user);

/*
 * Command reference (arguments in prhs order, results in plhs order):
 *   init, release                          -> no results                        (prost.cpp:278-281)
 *   list_gpus                              -> prints one line per device        (:283-297)
 *   set_gpu(id)                                                                 (:299-303)
 *   solve_problem(problem, nrows, ncols, backend, opts) -> struct {x,y,z,w,result[,iters,path,pair_launches]}  (:68-155)
 *       problem: struct {linop, prox_g, prox_f, prox_gstar, prox_fstar, scaling, scaling_alpha |
 *                        scaling_left, scaling_right}                           (factory.cpp:950-990)
 *       backend: cell {name, struct}   name in {pdhg, admm}                     (:914-948)
 *       opts:    struct of prost.options                                        (:992-1012)
 *   eval_linop(linop_cells, rhs, transpose) -> result, rowsum, colsum, time_ms  (prost.cpp:157-224)
 *   eval_prox(prox_cell, arg, tau, Tau[, verbose]) -> result, time_ms           (:226-276)
 *   set_precision('single'|'double'), get_precision -> string
 *   problem_info(problem, nrows, ncols) -> struct {scaling_left, scaling_right, nrows, ncols,
 *                                                   prox_g, prox_f, prox_gstar, prox_fstar}  (index/size/name rows)
 *   glibc_rand_unit(n[, skip[, piece]]) -> n x 1 (drawn in consecutive pieces of `piece` values when given): (T)rand() / (T)RAND_MAX of a fresh process after `skip` draws (the start vector
 *       of Problem::normest, problem.cu:441-444; host only)
 *   solver_create(problem, nrows, ncols, backend, opts[, [x0 x1 nx]]) -> handle (scalar); the optional
 *       1x3 matrix marks image columns [x0, x1) of nx as OWNED (column-sharded images: the rest are halo
 *       columns that do not count in the residual sums); pdhg one-kernel gradient2d paths (L <= 4 channels) only
 *   solver_kernel_times(handle) -> the `kernels` cell of solver_iterate for the events recorded since the last evaluation
 *   solver_iterate(handle, iters[, time_kernels[, sample_every[, checked[, defer_times]]]]) -> struct {ms, converged, kernels};
 *       kernels = cell of {name, avg_ms, sampled launches, iterations per launch, all launches, chunk columns};
 *       sample_every: one launch in that many is bracketed by events (default 8; 1 = every launch);
 *       checked != 0 runs the loop of solve_problem (stopping test of solver.cu:141-150 after every observable
 *       iteration, stops when it fires) without callbacks or solution read-out; defer_times != 0 leaves `kernels`
 *       empty and the recorded events to solver_kernel_times
 *   solver_halo_exchange(handle, ny, halo, left_halo, right_halo, left_rank, right_rank): swap `halo` image
 *       columns of x and y with the neighbouring ranks over the comm_init communicator (rank < 0: none)
 *   solver_iterate_sharded(handle, iters, ny, halo, left_halo, right_halo, left_rank, right_rank, since_exchange) ->
 *       since_exchange: `iters` iterations of a column slab with the halo refresh every halo - 2 iterations, the loop
 *       inside the native solver (collective over the communicator)
 *   solver_copy_columns(dst_handle, dst_col, src_handle, src_col, ncols, ny): the same transfer between two
 *       solvers of one process
 *   solver_state(handle[, with_vectors = 1]) -> struct {x,y,z,w,tau,sigma,theta,rho,primal_res,dual_res,primal_var_norm,
 *                                   dual_var_norm,eps_primal,eps_dual,iteration,cg_iterations,path,pair_launches,
 *                                   speculative_launches,speculative_adopted,sparse_pattern_products}; with_vectors = 0 leaves out x,y,z,w
 *   solver_compare(handle_a, handle_b) -> 4x2 matrix, rows x, y, x_prev, y_prev: {elements that differ in value,
 *       sum |a - b|}, formed on the device (verification of states too large to read back; pdhg only)
 *   solver_read(handle, 'x'|'y'|'x_prev'|'y_prev', offsets, count) -> count x numel(offsets) matrix: `count`
 *       consecutive entries of that device vector from every offset (partial read-back; pdhg only)
 *   solver_destroy(handle)
 *   comm_unique_id -> 1x128 matrix of byte values;  comm_init(id, rank, world);  comm_destroy
 *   comm_info -> struct {nranks, transport}: ranks as the communicator counts them (ncclCommCount), 'rccl' | 'host' | 'none'
 *   load_plugin(path): dlopen a shared library of user-defined prost::Block / Prox / Backend subclasses that register
 *       themselves in Factory<T>::block_reg() / prox_reg() / backend_reg() from static initialisers (custom.cpp:11-28;
 *       the reference compiles such sources into the MEX file, cmake/CustomSources.cmake.example:1-26)
 *   registered -> struct {prox, block, backend}: cells of the registered names
 *   set_quirks(struct {diags_adjoint_grid, dual_negate_float, fuse_moreau, sparse_patterns, sparse_stencils}) -- the first two switch reference
 *       bug-compatibility on; the others (default on) switch MI355X-side fusions off for A/B runs (sparse_stencils: a sparse block that IS
 *       spmat_gradient2d(nx, ny, 1) runs the fused gradient kernels with the preconditioners of the matrix)
 */

#ifdef __cplusplus
}
#endif
#endif /* PROST_C_H_ */
