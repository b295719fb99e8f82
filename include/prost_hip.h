/*
 * prost_hip.h -- thin C ABI of the MI355X (gfx950) kernels behind the prost hot path.
 *
 * This is the drop-in boundary of the build: plain pointers and sizes, no C++
 * types, no exceptions, no hidden synchronisation.  Every entry point
 *   - returns 0 on success, otherwise a non-zero code; the message is prost_hip_last_error();
 *   - takes DEVICE pointers unless a parameter is documented as host;
 *   - enqueues on `stream` (a hipStream_t passed as void*; NULL = the null stream) and returns
 *     without waiting.
 * Suffix _f32 / _f64 selects the arithmetic type T (the reference instantiates both:
 * `template class X<float>; template class X<double>;`).
 *
 * Each declaration cites the reference interface (file:line under /root/reference) it replaces.
 * The host C++ layer (include/prost/ *.hpp, libprost.so) calls only these functions; a cgo /
 * MEX / ctypes binding would bind exactly these symbols.
 */
#ifndef PROST_HIP_H_
#define PROST_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: prost_hip_fused_desc gained res_x0 / res_x1; fused_iteration2, comm send/recv, wrapper proxes, Kronecker blocks
 * 3: additions only -- device-resident CGLS / ADMM stages, prox_elem_arg, csr_spmv (non-accumulating), fused_iteration3d,
 *    fused_iteration3d_pw, fused_iteration3d_x2, fused_iteration_mc, fused_iteration_mc_x2, normest_grad_round, stream_wait_event, graph capture
 * 4: prost_hip_selftest_math writes EIGHT counters (it wrote five up to an early v3 header: a caller built against that
 *    header passes a too short buffer -- check prost_hip_abi_version() >= 4 before relying on the 8-slot layout);
 *    additions: comm_count, comm_is_host, comm_host_configure (point-to-point on the host-callback transport),
 *    fused operator entry points of the ADMM graph projection, mask_merge, next_launch_events, pattern_spmv
 * 5: additions only -- device-resident step-size rules (pdhg_rule_*, fused_iteration_rec, fused_iteration2_rec)
 * 6: prost_hip_fused_desc gained f_moreau (appended); additions: event_create_timing; next_launch_events takes (NULL, stop)
 * 7: prost_hip_op_block gained the row-pattern fields, prost_hip_arg_spec the operator source (both appended; callers that filled the
 *    old layouts must zero the new members); prost_hip_cgls_workspace_bytes doubled (order-independent sums: (hi, lo) per partial);
 *    additions: cgls_pixel_round / _close, pixel_op_supported, fused_iteration3d_rec / _3d_pw_rec / _3d_x2_rec, pdhg_fold_sums
 * 8: prost_hip_op_block gained anchor / anchor_t (appended; they must be NULL unless ids / ids_t are set -- callers that filled the old
 *    layout must zero them); additions: pattern_spmv_anchored_*, pdhg_record_view, the non-accumulating Kronecker entry points
 * 9: prost_hip_fused_desc gained arith (appended; callers that filled the old layout must zero it = PROST_HIP_ARITH_EXACT);
 *    additions: fused_iteration2_arith, fused_iteration_mc_x2_arith, fused_iteration3d_x2_arith, fused_iterationk_* (K iterations per launch)
 * 10: prost_hip_pixel_op gained d_csr and the CSR arrays of D and D^T (appended; callers that filled the old layout must zero them) */
#define PROST_HIP_ABI_VERSION 10

/* ------------------------------------------------------------------------------------------ */
/* runtime plumbing (replaces cudaSetDevice/cudaDeviceReset/thrust::device_vector allocation:  */
/* matlab/+prost/private/prost.cpp:69-72,:299-303; src/backend/backend_pdhg.cu:210-221)        */
/* ------------------------------------------------------------------------------------------ */
const char* prost_hip_last_error(void);
int prost_hip_abi_version(void);
int prost_hip_device_count(int* count);
int prost_hip_set_device(int device);
int prost_hip_get_device(int* device);
/* name: host buffer of `len` bytes; cu_count/total_mem may be NULL (prost.cpp:84-89, :283-297) */
int prost_hip_device_info(int device, char* name, size_t len, int* cu_count, size_t* total_mem);
int prost_hip_mem_info(size_t* free_bytes, size_t* total_bytes);          /* solver.cu:103 */
int prost_hip_malloc(void** dptr, size_t bytes);
int prost_hip_free(void* dptr);
int prost_hip_host_alloc(void** hptr, size_t bytes);                      /* pinned host memory */
int prost_hip_host_free(void* hptr);
int prost_hip_memcpy_h2d(void* dst, const void* src_host, size_t bytes, void* stream);
int prost_hip_memcpy_d2h(void* dst_host, const void* src, size_t bytes, void* stream);
int prost_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int prost_hip_memset(void* dst, int value, size_t bytes, void* stream);
int prost_hip_stream_create(void** stream);
int prost_hip_stream_destroy(void* stream);
int prost_hip_stream_synchronize(void* stream);
int prost_hip_device_synchronize(void);
/* Kernel timing: the NEXT iteration-kernel launch of the calling thread (the fused PDHG iteration kernels and the kernels of
 * prost_hip_cgls_round) takes `start` / `stop` (events of prost_hip_event_create / _create_timing) into hipExtLaunchKernel.  `stop` is
 * bound to the kernel's own command (its end; free); `start` is a marker packet of its own in front of the kernel (~3.4-4.6 us lost in
 * a back-to-back chain), and prost_hip_event_elapsed_ms(start, stop) is the kernel's duration as a profiler reports it.  `start` may be
 * NULL: the distance between the stop events of two consecutive launches is then the launch PERIOD (kernel + the idle time between
 * dependent launches).  (NULL, NULL) withdraws events no launch has taken.  Launches that record nothing else (reduction folds,
 * generic kernels) do not take the events. */
int prost_hip_next_launch_events(void* start, void* stop);
/* Device-resident step sizes on the GENERIC PDHG path (ABI 6): while `record` (a record of prost_hip_pdhg_rule_begin) is set for the
 * calling thread, prost_hip_prox_elem_arg in the PDHG_PRIMAL / PDHG_DUAL modes, prost_hip_pdhg_primal_arg / _dual_arg and
 * prost_hip_pdhg_residual_primal / _dual read tau, sigma, theta from it ON THE DEVICE -- the values passed by the host are ignored --
 * and the first three return at once when its stop flag is raised.  With prost_hip_pdhg_rule_apply behind the residual sums a
 * batch of generic iterations runs without a host wait.  NULL withdraws it. */
int prost_hip_use_step_record(void* record);
/* HIP graphs: the launches enqueued on `stream` between begin and end are recorded instead of executed;
 * end returns an executable graph that replays them with one host call (launch-bound inner loops). */
int prost_hip_stream_begin_capture(void* stream);
int prost_hip_stream_end_capture(void* stream, void** graph_exec);
int prost_hip_graph_launch(void* graph_exec, void* stream);
int prost_hip_graph_destroy(void* graph_exec);
int prost_hip_event_create(void** event);
/* an event for kernel timing only (prost_hip_next_launch_events): created without the system-scope release a default event adds to the
 * end of the command it is bound to -- nothing the host reads from memory may be ordered by it */
int prost_hip_event_create_timing(void** event);
int prost_hip_event_destroy(void* event);
int prost_hip_event_record(void* event, void* stream);
int prost_hip_event_synchronize(void* event);
int prost_hip_stream_wait_event(void* stream, void* event);      /* later work on `stream` waits for `event` (no host wait) */
int prost_hip_event_elapsed_ms(void* start, void* stop, float* ms);
/* one hipGetLastError per iteration instead of a device sync per prox launch
 * (prox_elem_operation.inl:128-138) */
int prost_hip_check_last_error(void);

/* ------------------------------------------------------------------------------------------ */
/* linear operator blocks                                                                      */
/* ------------------------------------------------------------------------------------------ */
/* BlockGradient2DKernel / ...Adjoint (src/linop/block_gradient2d.cu:26-78, :81-139; launches
 * :166-204).  fwd: res[2*nx*ny*L] from rhs[nx*ny*L]; adj: res[nx*ny*L] from rhs[2*nx*ny*L].
 * acc != 0: res += K rhs (the reference's EvalLocalAdd semantics); acc == 0: res = K rhs
 * (saves the thrust::fill pass of linearoperator.cu:140-147 when a block owns its rows). */
int prost_hip_grad2d_fwd_f32(float* res, const float* rhs, size_t nx, size_t ny, size_t L, int label_first, int acc, void* stream);
int prost_hip_grad2d_fwd_f64(double* res, const double* rhs, size_t nx, size_t ny, size_t L, int label_first, int acc, void* stream);
int prost_hip_grad2d_adj_f32(float* res, const float* rhs, size_t nx, size_t ny, size_t L, int label_first, int acc, void* stream);
int prost_hip_grad2d_adj_f64(double* res, const double* rhs, size_t nx, size_t ny, size_t L, int label_first, int acc, void* stream);
/* BlockGradient3DKernel / ...Adjoint (src/linop/block_gradient3d.cu:25-81, :83-150; :177-248),
 * Dirichlet boundary at l = L-1 (:73-76).  fwd res has 3*nx*ny*L entries. */
int prost_hip_grad3d_fwd_f32(float* res, const float* rhs, size_t nx, size_t ny, size_t L, int label_first, int acc, void* stream);
int prost_hip_grad3d_fwd_f64(double* res, const double* rhs, size_t nx, size_t ny, size_t L, int label_first, int acc, void* stream);
int prost_hip_grad3d_adj_f32(float* res, const float* rhs, size_t nx, size_t ny, size_t L, int label_first, int acc, void* stream);
int prost_hip_grad3d_adj_f64(double* res, const double* rhs, size_t nx, size_t ny, size_t L, int label_first, int acc, void* stream);
/* BlockDiagsKernel / AdjointKernel (src/linop/block_diags.cu:36-96, launches :187-220).
 * offsets (sorted ascending, :99-119) and factors (float even for f64, :30,:108) are DEVICE
 * arrays of ndiags entries (replaces the __constant__ arrays :30-31).  Always accumulates.
 * ref_grid_quirk != 0 reproduces the adjoint grid sized from nrows (:211): columns
 * >= ceil(nrows/256)*256 are left untouched. */
int prost_hip_diags_fwd_f32(float* res, const float* rhs, size_t nrows, size_t ncols, size_t ndiags, const int64_t* offsets, const float* factors, void* stream);
int prost_hip_diags_fwd_f64(double* res, const double* rhs, size_t nrows, size_t ncols, size_t ndiags, const int64_t* offsets, const float* factors, void* stream);
int prost_hip_diags_adj_f32(float* res, const float* rhs, size_t nrows, size_t ncols, size_t ndiags, const int64_t* offsets, const float* factors, int ref_grid_quirk, void* stream);
int prost_hip_diags_adj_f64(double* res, const double* rhs, size_t nrows, size_t ncols, size_t ndiags, const int64_t* offsets, const float* factors, int ref_grid_quirk, void* stream);
/* res += A rhs, A in CSR with int32 indices: replaces cusparse{S,D}csrmv(alpha=beta=1) at
 * src/linop/block_sparse.cu:156-168 (forward, K) and :190-202 (adjoint, stored K^T). */
int prost_hip_csr_spmv_acc_f32(float* res, const float* rhs, size_t nrows, size_t nnz, const float* val, const int32_t* ptr, const int32_t* ind, void* stream);
int prost_hip_csr_spmv_acc_f64(double* res, const double* rhs, size_t nrows, size_t nnz, const double* val, const int32_t* ptr, const int32_t* ind, void* stream);
/* res = K rhs (non-accumulating form: the zero fill of Block::EvalLocal folded into the product) */
int prost_hip_csr_spmv_f32(float* res, const float* rhs, size_t nrows, size_t nnz, const float* val, const int32_t* ptr, const int32_t* ind, void* stream);
int prost_hip_csr_spmv_f64(double* res, const double* rhs, size_t nrows, size_t nnz, const double* val, const int32_t* ptr, const int32_t* ind, void* stream);
/* res (+)= A rhs for a matrix whose rows repeat a few (column - row, value) sequences -- stencils written out as sparse matrices
 * (spmat_gradient2d.m, blur operators): ids[row] (16-bit, the array 8-byte aligned and padded to a multiple of 4 rows) selects entries pptr[id] .. pptr[id + 1] - 1 of the
 * table (rel = column - row, pval = value), summed in that order: the bits of prost_hip_csr_spmv* for rows of <= 6 entries on
 * average.  2 bytes per row instead of 8 per entry + 4 per row.  acc = 1 accumulates (block_sparse.cu:156-168), 0 writes. */
int prost_hip_pattern_spmv_f32(float* res, const float* rhs, size_t nrows, const uint16_t* ids, const int32_t* pptr, const int32_t* rel, const float* pval, int acc, void* stream);
int prost_hip_pattern_spmv_f64(double* res, const double* rhs, size_t nrows, const uint16_t* ids, const int32_t* pptr, const int32_t* rel, const double* pval, int acc, void* stream);
/* (ABI 6) the same with the table's sizes (npatterns + 1 offsets in pptr, nentries entries): a table of <= 256 patterns and <= 1024
 * entries is staged in LDS by every workgroup -- two dependent memory round trips less per wavefront, which is what small products
 * (<= 2^22 rows) consist of; same arithmetic, same bits */
int prost_hip_pattern_spmv_tab_f32(float* res, const float* rhs, size_t nrows, const uint16_t* ids, const int32_t* pptr, const int32_t* rel,
    const float* pval, int npatterns, int nentries, int acc, void* stream);
int prost_hip_pattern_spmv_tab_f64(double* res, const double* rhs, size_t nrows, const uint16_t* ids, const int32_t* pptr, const int32_t* rel,
    const double* pval, int npatterns, int nentries, int acc, void* stream);
/* (ABI 8) ANCHORED row patterns: the table's offsets count from anchor[row] (int32, padded like ids to a multiple of 4 rows, 16-byte aligned) instead
 * of from the row number: rows of equal SHAPE share a pattern also where the matrix maps between different geometries -- convmtx2's FULL
 * convolution in example_deblurring.m:15-16 (the blurred image is larger than the sharp one: column - row drifts by ky - 1 per image
 * column), restrictions / prolongations, rectangular crops.  Entry k of row r multiplies rhs[anchor[r] + rel[k]].  6 bytes per row instead
 * of 8 per entry.  Same sums in the same order as prost_hip_csr_spmv with one lane per row. */
int prost_hip_pattern_spmv_anchored_f32(float* res, const float* rhs, size_t nrows, const uint16_t* ids, const int32_t* anchor, const int32_t* pptr, const int32_t* rel,
                                        const float* pval, int npatterns, int nentries, int acc, void* stream);
int prost_hip_pattern_spmv_anchored_f64(double* res, const double* rhs, size_t nrows, const uint16_t* ids, const int32_t* anchor, const int32_t* pptr, const int32_t* rel,
                                        const double* pval, int npatterns, int nentries, int acc, void* stream);
/* res += kron(K, I_d) rhs (BlockSparseKronIdKernel, src/linop/block_sparse_kron_id.cu:26-49) and
 * res += kron(I_d, K) rhs (BlockIdKronSparseKernel, src/linop/block_id_kron_sparse.cu:26-52); K (nrows x ncols)
 * in CSR with int32 indices and FLOAT values for both T (:36, :79).  The adjoint is the same call
 * on the stored transpose. */
int prost_hip_sparse_kron_id_acc_f32(float* res, const float* rhs, size_t diaglength, size_t nrows, const float* val, const int32_t* ptr, const int32_t* ind, void* stream);
int prost_hip_sparse_kron_id_acc_f64(double* res, const double* rhs, size_t diaglength, size_t nrows, const float* val, const int32_t* ptr, const int32_t* ind, void* stream);
int prost_hip_id_kron_sparse_acc_f32(float* res, const float* rhs, size_t diaglength, size_t nrows, size_t ncols, const float* val, const int32_t* ptr, const int32_t* ind, void* stream);
int prost_hip_id_kron_sparse_acc_f64(double* res, const double* rhs, size_t diaglength, size_t nrows, size_t ncols, const float* val, const int32_t* ptr, const int32_t* ind, void* stream);
/* (ABI 8) res = ... : the non-accumulating forms (the zero fill of Block::EvalLocal folded into the product) */
int prost_hip_sparse_kron_id_f32(float* res, const float* rhs, size_t diaglength, size_t nrows, const float* val, const int32_t* ptr, const int32_t* ind, void* stream);
int prost_hip_sparse_kron_id_f64(double* res, const double* rhs, size_t diaglength, size_t nrows, const float* val, const int32_t* ptr, const int32_t* ind, void* stream);
int prost_hip_id_kron_sparse_f32(float* res, const float* rhs, size_t diaglength, size_t nrows, size_t ncols, const float* val, const int32_t* ptr, const int32_t* ind, void* stream);
int prost_hip_id_kron_sparse_f64(double* res, const double* rhs, size_t diaglength, size_t nrows, size_t ncols, const float* val, const int32_t* ptr, const int32_t* ind, void* stream);
/* x = beta * x (beta == 0 -> zero fill): thrust::fill / transform at linearoperator.cu:140-147 */
/* x[i] = value (thrust::fill, linearoperator.cu:140-147; also used for preconditioners that are one constant: the
 * gradient blocks' row / column sums are, block_gradient2d.cu:154-163, so nothing is uploaded for them) */
int prost_hip_fill_f32(float* x, double value, size_t n, void* stream);
int prost_hip_fill_f64(double* x, double value, size_t n, void* stream);
int prost_hip_scale_f32(float* x, size_t n, double beta, void* stream);
int prost_hip_scale_f64(double* x, size_t n, double beta, void* stream);
/* x = -x, optionally through a float round trip: thrust::negate<float> at
 * dual_linearoperator.cu:56-57,:78-79 (via_float reproduces it for f64) */
int prost_hip_negate_f32(float* x, size_t n, void* stream);
int prost_hip_negate_f64(double* x, size_t n, int via_float, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* proximal operators                                                                          */
/* ------------------------------------------------------------------------------------------ */
/* function ids, in the order of the registry names of factory.cpp:21-48 */
enum {
  PROST_FN_ZERO = 0, PROST_FN_ABS, PROST_FN_SQUARE, PROST_FN_IND_LEQ0, PROST_FN_IND_GEQ0,
  PROST_FN_IND_EQ0, PROST_FN_IND_BOX01, PROST_FN_MAX_POS0, PROST_FN_L0, PROST_FN_HUBER,
  PROST_FN_LQ, PROST_FN_LQ_PLUS_EPS, PROST_FN_TRUNCLIN, PROST_FN_TRUNCQUAD, PROST_FN_COUNT
};
enum { PROST_OP_1D = 0, PROST_OP_NORM2 = 1 };

/* ProxElemOperationKernel<T, ElemOperation1D|ElemOperationNorm2<T, Function1D*>>
 * (include/prost/prox/prox_elem_operation.inl:59-94, launch :153-187;
 * elem_operation_1d.hpp:36-59, elem_operation_norm2.hpp:40-88, function_1d.hpp:34-326).
 * coeff_ptr: HOST array of 7 DEVICE pointers (NULL entry = use scalar coeff_val[i]);
 * coeff_val: HOST array of 7 doubles (narrowed to T).  Layout: Vector index
 * interleaved ? tx*dim+i : tx+count*i (vector.hpp:44-48).  No device sync afterwards. */
int prost_hip_prox_elem_f32(int op, int fn, float* res, const float* arg, const float* tau_diag, double tau, int invert_tau,
                            size_t count, size_t dim, int interleaved, const float* const* coeff_ptr, const double* coeff_val, void* stream);
int prost_hip_prox_elem_f64(int op, int fn, double* res, const double* arg, const double* tau_diag, double tau, int invert_tau,
                            size_t count, size_t dim, int interleaved, const double* const* coeff_ptr, const double* coeff_val, void* stream);
/* The prox of the CONJUGATE of the same elem operation in one pass: ProxMoreau::EvalLocal (src/prox/prox_moreau.cu:98-134)
 * = MoreauPrescale (:29-43) + the elem operation with the inverted step flag + MoreauPostscale (:45-61), bit-identical
 * to prost_hip_moreau_prescale + prost_hip_prox_elem(!invert_tau) + prost_hip_moreau_postscale; 3 instead of 9 values
 * per element through HBM.  invert_tau is the flag the Moreau wrapper itself is called with. */
int prost_hip_prox_elem_moreau_f32(int op, int fn, float* res, const float* arg, const float* tau_diag, double tau, int invert_tau,
                                   size_t count, size_t dim, int interleaved, const float* const* coeff_ptr, const double* coeff_val, void* stream);
int prost_hip_prox_elem_moreau_f64(int op, int fn, double* res, const double* arg, const double* tau_diag, double tau, int invert_tau,
                                   size_t count, size_t dim, int interleaved, const double* const* coeff_ptr, const double* coeff_val, void* stream);
/* The same operations with the argument formed on the fly (no separate argument pass):
 *   PROST_ARG_PLAIN        arg = v[0]
 *   PROST_ARG_PDHG_PRIMAL  arg = v[0] - s[0] v[1] v[2]                          primal_proxarg_functor, backend_pdhg.cu:38-51
 *                          (x, T, K^T y; tau)
 *   PROST_ARG_PDHG_DUAL    arg = v[0] + s[0] v[1] ((1 + s[1]) v[2] - s[1] v[3])  dual_proxarg_functor, backend_pdhg.cu:54-70
 *                          (y, Sigma, K x, K x_prev; sigma, theta)
 * v[] are device pointers to the FIRST element of the prox's range in each vector; res must not alias v[0].
 * moreau != 0 evaluates the conjugate as prost_hip_prox_elem_moreau does. */
/* ABI 7, the operator sources: the same two arguments with the OPERATOR PRODUCT formed on the fly from the blocks of `op` (sparse blocks
 * as CSR arrays or row patterns, gradient stencils; evaluated per element in the order LinearOperator::Eval / EvalAdjoint accumulate the
 * blocks, linearoperator.cu:135-170 -- the bits of the separate passes), so K x and K^T y are never written:
 *   PROST_ARG_PDHG_PRIMAL_OP  arg = v[0] - s[0] v[1] (K^T w[0])                         v[0] = x, v[1] = T (prox range); w[0] = y (WHOLE dual vector);
 *                             v[3] = K^T y_prev of the range (read for the residual sums only), kty_out: K^T w[0] of the range is stored there
 *   PROST_ARG_PDHG_DUAL_OP    arg = v[0] + s[0] v[1] ((1 + s[1]) K w[0] - s[1] K w[1])  v[0] = y, v[1] = Sigma (prox range); w[0] = x, w[1] = x_prev (WHOLE)
 * base = index of the prox's first element in the whole variable; use[i] = 0 makes product i the zero vector (iterations 0 / 1 of the
 * reference, backend_pdhg.cu:213-216).  res_ws != NULL: the launch also adds up the terms of the dual (PRIMAL_OP) / primal (DUAL_OP) residual
 * of its elements (backend_pdhg.cu:73-120) -- one slot of 4 doubles per workgroup from slot res_slot on, at most res_slots_max workgroups;
 * prost_hip_pdhg_fold_sums folds the slots of an iteration's launches.  Planar layout (or the 1-D operation), count a multiple of 16 bytes of
 * elements, 16-byte aligned operands (prost_hip_prox_elem_arg_op_supported). */
typedef struct prost_hip_arg_spec {
  int mode; const void* v[4]; double s[2];
  const struct prost_hip_fused_op* op; uint64_t op_rows, op_cols; uint64_t base; const void* w[2]; void* kty_out; int use[2];
  double* res_ws; unsigned res_slot, res_slots_max;
} prost_hip_arg_spec;
enum { PROST_ARG_PLAIN = 0, PROST_ARG_PDHG_PRIMAL = 1, PROST_ARG_PDHG_DUAL = 2, PROST_ARG_PDHG_PRIMAL_OP = 3, PROST_ARG_PDHG_DUAL_OP = 4 };
/* 1 if the operator sources take `op` (m rows, n columns): sparse / gradient blocks whose positions and sizes -- and a gradient's height
 * and plane size -- are multiples of 16 bytes of elements (dtype 0: 4, 1: 2); host-only check */
int prost_hip_prox_elem_arg_op_supported(const struct prost_hip_fused_op* op, uint64_t m, uint64_t n, int dtype);
int prost_hip_prox_elem_arg_f32(int op, int fn, int moreau, float* res, const prost_hip_arg_spec* arg, const float* tau_diag, double tau, int invert_tau,
                                size_t count, size_t dim, int interleaved, const float* const* coeff_ptr, const double* coeff_val, void* stream);
int prost_hip_prox_elem_arg_f64(int op, int fn, int moreau, double* res, const prost_hip_arg_spec* arg, const double* tau_diag, double tau, int invert_tau,
                                size_t count, size_t dim, int interleaved, const double* const* coeff_ptr, const double* coeff_val, void* stream);
/* ProxIndEpiQuadKernel (src/prox/prox_ind_epi_quad.cu:42-79) + helper::ProjectEpiQuadNd
 * (include/prost/prox/helper.hpp:44-105).  a_ptr/c_ptr NULL -> scalar a_val/c_val. */
int prost_hip_prox_epi_quad_f32(float* res, const float* arg, size_t count, size_t dim, const float* a_ptr, double a_val, const float* b_ptr, const float* c_ptr, double c_val, void* stream);
int prost_hip_prox_epi_quad_f64(double* res, const double* arg, size_t count, size_t dim, const double* a_ptr, double a_val, const double* b_ptr, const double* c_ptr, double c_val, void* stream);
/* MoreauPrescale / MoreauPostscale (src/prox/prox_moreau.cu:29-61, used :110-133) */
int prost_hip_moreau_prescale_f32(float* scaled, const float* arg, const float* tau_diag, double tau, int invert_tau, size_t n, void* stream);
int prost_hip_moreau_prescale_f64(double* scaled, const double* arg, const double* tau_diag, double tau, int invert_tau, size_t n, void* stream);
int prost_hip_moreau_postscale_f32(float* res, const float* arg, const float* tau_diag, double tau, int invert_tau, size_t n, void* stream);
int prost_hip_moreau_postscale_f64(double* res, const double* arg, const double* tau_diag, double tau, int invert_tau, size_t n, void* stream);

/* ProxTransform (src/prox/prox_transform.cu), h(x) = c f(ax - b) + dx + (e/2) x^2 around any inner prox:
 * prescale = ProxTransformPrescaleArgument (:27-52) + ProxTransformPrescaleStepSize (:54-78) in one pass:
 *   tau2 = tau * tau_diag (or its reciprocal); scaled_arg = a (arg - tau2 d) / (1 + tau2 e) - b;
 *   scaled_tau = a^2 c tau2 / (1 + tau2 e).
 * coeff_ptr / coeff_val: a, b, c, d, e as in prost_hip_prox_elem (ptr NULL -> scalar).  The caller then
 * evaluates the inner prox with (scaled_arg, scaled_tau, tau = 1, invert = 0) and calls postscale
 * (:80-97): result = (result + b) / a. */
int prost_hip_transform_prescale_f32(float* scaled_arg, float* scaled_tau, const float* arg, const float* tau_diag, const float* const* coeff_ptr,
                                     const double* coeff_val, double tau, int invert_tau, size_t n, void* stream);
int prost_hip_transform_prescale_f64(double* scaled_arg, double* scaled_tau, const double* arg, const double* tau_diag, const double* const* coeff_ptr,
                                     const double* coeff_val, double tau, int invert_tau, size_t n, void* stream);
int prost_hip_transform_postscale_f32(float* result, const float* a_ptr, double a_val, const float* b_ptr, double b_val, size_t n, void* stream);
int prost_hip_transform_postscale_f64(double* result, const double* a_ptr, double a_val, const double* b_ptr, double b_val, size_t n, void* stream);
/* ProxPermuteKernel (src/prox/prox_permute.cu:31-48): inverse == 0: res[i] = arg[perm[i]]; else res[perm[i]] = arg[i] */
int prost_hip_permute_f32(float* res, const float* arg, const int32_t* perm, size_t n, int inverse, void* stream);
int prost_hip_permute_f64(double* res, const double* arg, const int32_t* perm, size_t n, int inverse, void* stream);
/* ProxIndHalfspaceKernel (src/prox/prox_ind_halfspace.cu:31-86): projection onto {x | <a, x> <= b} per group,
 * planar layout whatever `interleaved` says; sz_a = count*dim (planar per-group normals) or dim; sz_b = count or 1 */
int prost_hip_prox_ind_halfspace_f32(float* res, const float* arg, size_t count, size_t dim, const float* a, size_t sz_a, const float* b, size_t sz_b, void* stream);
int prost_hip_prox_ind_halfspace_f64(double* res, const double* arg, size_t count, size_t dim, const double* a, size_t sz_a, const double* b, size_t sz_b, void* stream);
/* ProxIndSOCKernel (src/prox/prox_ind_soc.cu:30-77): projection onto ||x|| <= y, (x_1 .. x_{dim-1}, y) planar */
int prost_hip_prox_ind_soc_f32(float* res, const float* arg, size_t count, size_t dim, void* stream);
int prost_hip_prox_ind_soc_f64(double* res, const double* arg, size_t count, size_t dim, void* stream);
/* ProxIndSumKernel (src/prox/prox_ind_sum.cu:30-66): sum over an index family == total_sum; `res` must
 * already hold a copy of arg (the reference's thrust::copy at :119); inds are local 64-bit indices */
int prost_hip_prox_ind_sum_f32(float* res, const float* arg, const float* tau_diag, const uint64_t* inds, size_t count, size_t dim, double total_sum, double tau, int invert_tau, void* stream);
int prost_hip_prox_ind_sum_f64(double* res, const double* arg, const double* tau_diag, const uint64_t* inds, size_t count, size_t dim, double total_sum, double tau, int invert_tau, void* stream);
/* ElemOperationIndSum (include/prost/prox/elemop/elem_operation_ind_sum.hpp:41-60): sum-to-one per group */
int prost_hip_prox_elem_ind_sum_f32(float* res, const float* arg, size_t count, size_t dim, int interleaved, void* stream);
int prost_hip_prox_elem_ind_sum_f64(double* res, const double* arg, size_t count, size_t dim, int interleaved, void* stream);
/* ElemOperationIndSimplex (include/prost/prox/elemop/elem_operation_ind_simplex.hpp:40-119): projection onto the
 * unit simplex per group.  `work` is a DEVICE scratch buffer of count*dim entries (the reference uses a
 * 1024-entry per-thread local array, i.e. dim <= 1024; no such limit here). */
int prost_hip_prox_elem_ind_simplex_f32(float* res, const float* arg, float* work, size_t count, size_t dim, int interleaved, void* stream);
int prost_hip_prox_elem_ind_simplex_f64(double* res, const double* arg, double* work, size_t count, size_t dim, int interleaved, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* PDHG building blocks, generic path (src/backend/backend_pdhg.cu)                            */
/* ------------------------------------------------------------------------------------------ */
/* primal_proxarg_functor (:38-51, used :317-331): temp = x - tau * T .* kty */
int prost_hip_pdhg_primal_arg_f32(float* temp, const float* x, const float* T, const float* kty, double tau, size_t n, void* stream);
int prost_hip_pdhg_primal_arg_f64(double* temp, const double* x, const double* T, const double* kty, double tau, size_t n, void* stream);
/* dual_proxarg_functor (:54-70, used :349-364): temp = y + sigma * S .* ((1+theta) kx - theta kx_prev) */
int prost_hip_pdhg_dual_arg_f32(float* temp, const float* y, const float* S, const float* kx, const float* kx_prev, double sigma, double theta, size_t m, void* stream);
int prost_hip_pdhg_dual_arg_f64(double* temp, const double* y, const double* S, const double* kx, const double* kx_prev, double sigma, double theta, size_t m, void* stream);
/* Residual reductions (primal_residual_transform :97-120, dual_residual_transform :73-94,
 * thrust::transform_reduce :392-431).  Two-stage deterministic reduction: block partials in
 * `workspace` (prost_hip_reduce_workspace_bytes() bytes), then one block folds them into
 * out2[0] = sum diff^2, out2[1] = sum var^2 (DEVICE doubles).  Terms are evaluated in T as the
 * reference does; the accumulation is in double (the reference's order is unspecified). */
size_t prost_hip_reduce_workspace_bytes(void);
int prost_hip_pdhg_residual_primal_f32(double* out2, const float* y_prev, const float* y, const float* S, const float* kx_prev, const float* kx,
    double sigma, double theta, size_t m, void* workspace, void* stream);
int prost_hip_pdhg_residual_primal_f64(double* out2, const double* y_prev, const double* y, const double* S, const double* kx_prev, const double* kx,
    double sigma, double theta, size_t m, void* workspace, void* stream);
int prost_hip_pdhg_residual_dual_f32(double* out2, const float* x_prev, const float* x, const float* T, const float* kty_prev, const float* kty, double tau, size_t n, void* workspace, void* stream);
int prost_hip_pdhg_residual_dual_f64(double* out2, const double* x_prev, const double* x, const double* T, const double* kty_prev, const double* kty,
    double tau, size_t n, void* workspace, void* stream);
/* compute_w_variable_functor / compute_z_variable_functor (:147-186, used :524-560) */
int prost_hip_pdhg_w_variable_f32(float* w, const float* x_prev, const float* x, const float* T, const float* kty_prev, double tau, size_t n, void* stream);
int prost_hip_pdhg_w_variable_f64(double* w, const double* x_prev, const double* x, const double* T, const double* kty_prev, double tau, size_t n, void* stream);
int prost_hip_pdhg_z_variable_f32(float* z, const float* y_prev, const float* y, const float* S, const float* kx, const float* kx_prev, double sigma, double theta, size_t m, void* stream);
int prost_hip_pdhg_z_variable_f64(double* z, const double* y_prev, const double* y, const double* S, const double* kx, const double* kx_prev, double sigma, double theta, size_t m, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Fused PDHG passes for K = one gradient2d/gradient3d block (label_first = 0),                */
/* prox_g = one elem_operation:1d:<fn>, prox_fstar = one elem_operation:norm2:<fn> over the    */
/* 2L (3 for gradient3d) planar gradient components, uniform preconditioners.                  */
/* One PerformIteration (backend_pdhg.cu:313-381) = primal pass + dual pass; the vectors kx,   */
/* kx_prev, kty, kty_prev and temp of the reference are never materialised:                    */
/*   primal pass: reads y (m), x (n), g-coefficient vectors; writes x_new (n)                  */
/*   dual pass:   reads y (m), x_new (n), x_old (n);          writes y_new (m)                 */
/* = 11 floats / pixel / iteration for ROF.                                     */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int is3d;                 /* 0: gradient2d (dual dim 2L), 1: gradient3d (dual dim 3)            */
  size_t nx, ny, L;
  int g_fn;                 /* PROST_FN_* of prox_g (elem_operation:1d)                           */
  const void* g_coeff_ptr[7]; double g_coeff_val[7];
  int f_fn;                 /* PROST_FN_* of prox_fstar (elem_operation:norm2)                    */
  const void* f_coeff_ptr[7]; double f_coeff_val[7];
  double T_val;             /* uniform squared preconditioners (problem.cu:262-287 give           */
  double S_val;             /*   Sigma = 1/2, Tau = 1/4 (2-D), 1/6 (3-D) for gradient blocks)     */
  size_t res_x0, res_x1;    /* residual sums only over image columns [res_x0, res_x1); res_x1 == 0: all  */
                            /* (column-sharded images: the halo columns of a slab are not counted;        */
                            /*  honoured by prost_hip_fused_iteration / _iteration2)                      */
  int g_b_masked;           /* (ABI 4) g_coeff_ptr[1] is a MERGED stream made by prost_hip_mask_merge: b where the coefficient a of  */
                            /* prox_g is 1, the mask sentinel where a is 0 -- a binary per-pixel a (the inpainting mask of           */
                            /* example_tv_inpaint.m:23) folded into the b stream; g_coeff_ptr[0] must be NULL, g_coeff_val[0] = 1.  */
                            /* Honoured by prost_hip_fused_iteration2 and prost_hip_fused_iteration_mc_x2 only.                      */
  int var_T;                /* (ABI 5) the primal preconditioner is NOT uniform: the operator was handed over as the sparse matrix    */
  double T_cls[3];          /* spmat_gradient2d(nx, ny, 1) (example_rof_primal.m:10, :28) and Tau_j = 1 / (column sum of |K|) as        */
                            /* problem.cu:262-287 forms it -- T_cls[0..2] for pixels with 2, 3, 4 stencil entries in their column         */
                            /* (corner, edge, interior; count = 4 - [x == 0] - [x == nx - 1] - [y == 0] - [y == ny - 1]), T_val = T_cls[2]. */
                            /* Sigma stays uniform (the all-zero rows of the matrix inherit 1/2, problem.cu:267-286).  Honoured by        */
                            /* prost_hip_fused_iteration (L <= 2), _iteration2 (L == 1, ROF shape), _iteration_mc (L = 3, 4), _iteration_mc_x2 (ROF shape); the others refuse it. */
  int f_moreau;             /* (ABI 6) prox_fstar is the Moreau wrap of the described elem_operation:norm2 (a problem written in the PRIMAL  */
                            /* form, example_rof_primal.m:27: backend_pdhg.cu:255-266 derives prox_f* from prox_f; prox_moreau.cu:98-134):   */
                            /* v = arg / (sigma Sigma), r = prox of the described function at v with the step 1 / (sigma Sigma),             */
                            /* result = arg - sigma Sigma r.  Honoured by prost_hip_fused_iteration, _iteration_mc and (square data term with per-pixel b,        */
                            /* 'abs') _iteration2; the others refuse it. */
  int arith;                /* (ABI 9) PROST_HIP_ARITH_*: the arithmetic class the iteration kernels may use.  EXACT (0): every result    */
                            /* rounds like the reference's expressions evaluated WITHOUT contraction (the CPU oracle, bit for bit).        */
                            /* FMAD (1): fused multiply-adds where a product feeds a sum (nvcc's default for the reference's kernels,      */
                            /* src/CMakeLists.txt:12-24), quotients by 1 + step and by ||v|| through fp32 reciprocal / reciprocal square   */
                            /* root instructions -- results within the tolerance of tests/test_gpu_fmad.py of the exact ones.  Honoured by */
                            /* prost_hip_fused_iteration2, _iteration_mc_x2 and _iteration3d_x2 for fp32 ROF / TV-L1 shapes (see           */
                            /* prost_hip_fused_iteration2_arith); every other entry point and shape computes exactly.                      */
} prost_hip_fused_desc;
#define PROST_HIP_ARITH_EXACT 0
#define PROST_HIP_ARITH_FMAD 1
/* Folds a BINARY per-element coefficient a of ElemOperation1D (elem_operation_1d.hpp:42-44: a == 0 skips the function, the
 * element passes through) into the b stream: bm[i] = a[i] == 0 ? sentinel : (b ? b[i] : b_val), sentinel = a quiet NaN with the
 * payload 0xA5A5.. that arithmetic never produces (PROST_HIP_MASK_SENTINEL_*).  nonbinary (DEVICE counter, zeroed by the caller)
 * is incremented for every a[i] outside {0, 1}: the merged stream is only valid when it stays 0. */
#define PROST_HIP_MASK_SENTINEL_F32 0x7FC0A5A5u
#define PROST_HIP_MASK_SENTINEL_F64 0x7FF8A5A5A5A5A5A5ull
int prost_hip_mask_merge_f32(float* bm, const float* a, const float* b, double b_val, size_t n, unsigned long long* nonbinary, void* stream);
int prost_hip_mask_merge_f64(double* bm, const double* a, const double* b, double b_val, size_t n, unsigned long long* nonbinary, void* stream);

/* returns 1 if the fused passes support this description for dtype (0 f32, 1 f64) */
int prost_hip_fused_supported(const prost_hip_fused_desc* desc, int dtype);

/* primal pass of iteration k:  x_new = prox_g(x - tau T K^T y ; T, tau)            (:317-338)
 * use_kty == 0 reproduces iteration 0, where the reference's kty_ is the zero vector it was
 * allocated with, not K^T y0 (:213,:377-380).
 * res_out2 != NULL additionally accumulates the dual residual sums (:73-94, :413-431) with
 * kty = K^T y (if use_kty) and kty_prev = K^T y_prev (if use_kty_prev); needs `workspace`. */
int prost_hip_fused_primal_f32(const prost_hip_fused_desc* desc, float* x_new, const float* x, const float* y, const float* y_prev,
                               double tau, int use_kty, int use_kty_prev, double* res_out2, void* workspace, void* stream);
int prost_hip_fused_primal_f64(const prost_hip_fused_desc* desc, double* x_new, const double* x, const double* y, const double* y_prev,
                               double tau, int use_kty, int use_kty_prev, double* res_out2, void* workspace, void* stream);
/* dual pass of iteration k: y_new = prox_fstar(y + sigma S ((1+theta) K x_new - theta K x_old))
 * (:341-370).  use_kx_prev == 0 reproduces iteration 0 (kx_prev_ = zero vector, :216).
 * res_out2 != NULL additionally accumulates the primal residual sums (:97-120, :392-410). */
int prost_hip_fused_dual_f32(const prost_hip_fused_desc* desc, float* y_new, const float* y, const float* x_new, const float* x_old,
                             double sigma, double theta, int use_kx_prev, double* res_out2, void* workspace, void* stream);
int prost_hip_fused_dual_f64(const prost_hip_fused_desc* desc, double* y_new, const double* y, const double* x_new, const double* x_old,
                             double sigma, double theta, int use_kx_prev, double* res_out2, void* workspace, void* stream);

/* One WHOLE PerformIteration (:313-381) in a single kernel: a wavefront computes x_new one column
 * ahead of y_new in registers, so x_new never makes the HBM round trip between the two passes:
 * reads y (m), x (n), g-coefficients; writes x_new (n), y_new (m) = 7 floats / pixel / iteration
 * for ROF instead of 11.  Results are bit-identical to prost_hip_fused_primal + prost_hip_fused_dual.
 * cols_per_block <= 0 picks the column chunk.  res_out4 != NULL (residual iterations) also streams
 * y_prev and writes the four sums of :392-431 to res_out4 = {primal diff^2, primal var^2,
 * dual diff^2, dual var^2} (DEVICE doubles; needs `workspace`); use_kty_prev as in fused_primal. */
int prost_hip_fused_iteration_supported(const prost_hip_fused_desc* desc, int dtype);
int prost_hip_fused_iteration_f32(const prost_hip_fused_desc* desc, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev,
                                  double tau, double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols_per_block,
                                  double* res_out4, void* workspace, void* stream);
int prost_hip_fused_iteration_f64(const prost_hip_fused_desc* desc, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev,
                                  double tau, double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols_per_block,
                                  double* res_out4, void* workspace, void* stream);

/* TWO consecutive PerformIteration calls (:313-381, iterations k and k+1, both with k >= 2 and
 * neither a residual iteration) in ONE kernel: x^(k+1), y^(k+1) stay in registers, so the launch
 * reads x^k, y^k, g-coefficients and writes x^(k+2), y^(k+2) -- 7 floats / pixel per two
 * iterations.  tau/sigma/theta are HOST arrays of 2 (the step sizes of iteration k and k+1; alg2
 * changes them every iteration, :483-488).  Bit-identical to two prost_hip_fused_iteration
 * launches.  x_mid / y_mid == NULL: the intermediate iterate x^(k+1), y^(k+1) is not written
 * anywhere, callers must not need it.  x_mid, y_mid != NULL: it is stored there (10 floats/pixel),
 * leaving the same observable state as two single launches.  res_out4 != NULL (needs `workspace`):
 * also the four residual sums of iteration k+1 as prost_hip_fused_iteration writes them (same
 * terms; the summation order differs, so the sums agree to rounding) -- with or without x_mid.
 * gradient2d with L == 1 only; see _supported. */
int prost_hip_fused_iteration2_supported(const prost_hip_fused_desc* desc, int dtype);
/* 1 iff the launch is also FASTER than two prost_hip_fused_iteration launches: today the ROF / TV-L1 shapes (prox_g
 * square or abs with scalar a = 1, d = e = 0, prox_f* ind_leq0 with scalar a = 1, d = e = 0), which has a straight-line
 * instance; other function pairs are supported but register-bound (about 5x slower), so callers pair only here */
int prost_hip_fused_iteration2_profitable(const prost_hip_fused_desc* desc, int dtype);
/* columns per wavefront (chunk length) a launch with cols_per_block <= 0 uses for this description; 0 if unsupported.
 * Measurement key: the HBM traffic of a launch depends on it (3 warm-up columns are re-read per chunk). */
int prost_hip_fused_iteration2_chunk_cols(const prost_hip_fused_desc* desc, int dtype, int with_residuals);
/* (ABI 9) PROST_HIP_ARITH_* of the instance a launch with this description runs: FMAD only where desc->arith asks for it AND a
 * tolerance-class instance exists (fp32, straight-line ROF / TV-L1 shape, uniform Tau, ind_leq0 radius > 0), EXACT otherwise */
int prost_hip_fused_iteration2_arith(const prost_hip_fused_desc* desc, int dtype);

/* (ABI 9) K consecutive PerformIteration calls (:313-381; K = 1 .. 6, none of them observed from outside and none but the LAST a
 * residual iteration) in ONE kernel, tolerance-class arithmetic only (desc->arith == PROST_HIP_ARITH_FMAD): the iterates in between
 * stay in registers, the launch reads x^k, y^k and b once and writes x^(k+K), y^(k+K) -- 7 floats / pixel per K iterations.
 * tau / sigma / theta: HOST arrays of K.  res_out4 != NULL: the four residual sums of the last iteration.  gradient2d, L == 1, fp32,
 * heights that are a multiple of 4, prox_g = 1d:square | 1d:abs with scalar a = 1, d = e = 0, prox_f* = norm2:ind_leq0 with scalar
 * a = 1, b > 0, d = e = 0.  A launch equals the corresponding sequence of prost_hip_fused_iteration2 launches of the same arithmetic
 * class bit for bit.  _max: the largest K a description supports (0: none); _rec: step sizes from the device record (ABI 5). */
int prost_hip_fused_iterationk_max(const prost_hip_fused_desc* desc, int dtype);
int prost_hip_fused_iterationk_chunk_cols(const prost_hip_fused_desc* desc, int dtype, int k, int with_residuals);
int prost_hip_fused_iterationk_f32(const prost_hip_fused_desc* desc, int k, float* x_out, float* y_out, const float* x, const float* y,
                                   const double* tau, const double* sigma, const double* theta, int cols_per_block, double* res_out4,
                                   void* workspace, void* stream);
int prost_hip_fused_iterationk_f64(const prost_hip_fused_desc* desc, int k, double* x_out, double* y_out, const double* x, const double* y,
                                   const double* tau, const double* sigma, const double* theta, int cols_per_block, double* res_out4,
                                   void* workspace, void* stream);      /* always fails: fp32 only (_max answers 0) */

/* ---- device-resident step sizes for the residual-driven rules (ABI 5; kernels_pdhg_rule.hip) --------------------------------
 * Replaces the host side of BackendPDHG::UpdateResidualsAndStepsizes (backend_pdhg.cu:433-476: sqrt of the four sums, eps_primal /
 * eps_dual of backend.hpp:71-74, Goldstein's rule :443-460, Boyd's rule :462-476) and the stopping test of Solver::Solve
 * (solver.cu:141-150) by a one-thread kernel behind the reduction of the sums: same arithmetic, same precision (T = the _f32 /
 * _f64 suffix), same order.  `record` is DEVICE memory of prost_hip_pdhg_rule_record_bytes() bytes, owned by the caller.
 *   _begin : (re)initialises the record from the host's state -- the rule options, the description's scalar prox coefficients, tau,
 *            sigma, theta, Goldstein's alpha, Boyd's l / u -- and clears the stop word.  stop_on_convergence != 0: a rule
 *            evaluation whose stopping test fires raises the stop word; every prost_hip_fused_iteration*_rec launch and rule
 *            evaluation enqueued after it returns at once (the iterate stays that of the stopping iteration).
 *   _apply : one evaluation on sums4 = {primal diff^2, primal var^2, dual diff^2, dual var^2} (device or pinned host doubles, as
 *            prost_hip_fused_iteration* and the all-reduce leave them) for the residual iteration with index `iteration`
 *            (BackendPDHG::iteration_ when the reference calls UpdateResidualsAndStepsizes).
 *   mirror : optional struct (device memory, or pinned host memory the device can write) that receives every scalar after the
 *            evaluation; valid once the stream has been synchronised (and the struct copied, if it lives on the device).  Nothing
 *            here waits for the device.
 * prost_hip_fused_iteration_rec / _iteration2_rec: the launches of prost_hip_fused_iteration / _iteration2 with tau, sigma, theta
 * (and the prox terms derived from them) read from the record by the kernel instead of passed by value; both iterations of a
 * double-iteration launch use the record's current values.  Need scalar e = 0 on both proxes for the straight-line instances. */
#define PROST_PDHG_RULE_NONE 0        /* alg1: nothing to adapt, residuals / stopping test only */
#define PROST_PDHG_RULE_GOLDSTEIN 2
#define PROST_PDHG_RULE_BOYD 3
typedef struct prost_hip_pdhg_rule_opts {
  int variant;                                  /* PROST_PDHG_RULE_* */
  double arg_nu, arg_delta, arb_delta, arb_tau; /* pdhg.m:4-14 */
  double tol_abs_primal, tol_abs_dual, tol_rel_primal, tol_rel_dual;   /* options.m */
  double sqrt_rows, sqrt_cols;                  /* sqrt of the (global) numbers of dual / primal variables, formed in double on the host */
} prost_hip_pdhg_rule_opts;
typedef struct prost_hip_pdhg_rule_state {
  double tau, sigma, theta;                     /* after the last evaluation */
  double prev_tau, prev_sigma, prev_theta;      /* before it */
  double arg_alpha;
  long long arb_l, arb_u;
  double sums[4];
  double primal_res, dual_res, primal_var, dual_var, eps_primal, eps_dual;
  unsigned long long evaluations;               /* since _begin */
  unsigned long long last_iteration;            /* `iteration` of the last evaluation */
  long long stopped;                            /* the stopping test fired (stop_on_convergence) ... */
  unsigned long long stop_iteration;            /* ... at the residual iteration with this index */
} prost_hip_pdhg_rule_state;
/* Both residual reductions of an iteration in ONE launch and their folds -- with the rule and the stopping test of `record` behind
 * them when apply_rule is set -- in a second: sums4 = {primal diff^2, primal var^2, dual diff^2, dual var^2}, bit for bit what
 * prost_hip_pdhg_residual_primal + _dual give (same partials, same fold order).  record may be NULL when apply_rule is 0. */
int prost_hip_pdhg_residuals_f32(double* sums4, const float* y_prev, const float* y, const float* S, const float* kx_prev, const float* kx, double sigma, double theta, size_t m,
                                 const float* x_prev, const float* x, const float* T, const float* kty_prev, const float* kty, double tau, size_t n, void* workspace, void* record,
                                 int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_pdhg_residuals_f64(double* sums4, const double* y_prev, const double* y, const double* S, const double* kx_prev, const double* kx, double sigma, double theta, size_t m,
                                 const double* x_prev, const double* x, const double* T, const double* kty_prev, const double* kty, double tau, size_t n, void* workspace, void* record,
                                 int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
size_t prost_hip_pdhg_rule_record_bytes(void);
/* ABI 8: the DEVICE addresses of the scalars a record of prost_hip_pdhg_rule_begin_f32 / _f64 (dtype 0 / 1) holds for the coming iterations --
 * tau, sigma, theta (float or double) and the stop word (int, non-zero once the stopping test has fired) -- so that a kernel outside this
 * library (a plugin prox, include/prost/prox/prox.hpp: StepView) can take its step size from the device inside a batch of iterations.
 * Address arithmetic only: nothing is read or launched.  Any out pointer may be NULL. */
int prost_hip_pdhg_record_view(const void* record, int dtype, const void** tau, const void** sigma, const void** theta, const int** stop);
int prost_hip_pdhg_rule_begin_f32(void* record, const prost_hip_pdhg_rule_opts* opts, const prost_hip_fused_desc* desc, double tau, double sigma, double theta,
                                  double arg_alpha, int arb_l, int arb_u, int stop_on_convergence, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_pdhg_rule_begin_f64(void* record, const prost_hip_pdhg_rule_opts* opts, const prost_hip_fused_desc* desc, double tau, double sigma, double theta,
                                  double arg_alpha, int arb_l, int arb_u, int stop_on_convergence, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_pdhg_rule_apply_f32(void* record, const double* sums4, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_pdhg_rule_apply_f64(void* record, const double* sums4, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
/* The residual sums of an iteration whose prox launches added them up themselves (prost_hip_prox_elem_arg with an operator source and
 * res_ws): out4 = {primal: sum diff^2, sum z_hat^2 ; dual: sum diff^2, sum w_hat^2} (backend_pdhg.cu:392-431) from n_primal / n_dual slots of 4
 * doubles (both arrays on 32-byte boundaries); record != NULL and apply_rule: the step-size rule and the stopping test follow in the same
 * launch (prost_hip_pdhg_rule_apply). */
int prost_hip_pdhg_fold_sums_f32(double* out4, const double* ws_primal, unsigned n_primal, const double* ws_dual, unsigned n_dual, void* record, int apply_rule,
                                 unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_pdhg_fold_sums_f64(double* out4, const double* ws_primal, unsigned n_primal, const double* ws_dual, unsigned n_dual, void* record, int apply_rule,
                                 unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
/* apply_rule != 0 (res_out4 != NULL): the kernel that folds the four sums also evaluates the rule on them -- what _apply does, one launch
 * less per residual iteration; for residual iterations whose sums need no all-reduce.  `iteration`, `mirror`: as for _apply. */
int prost_hip_fused_iteration_rec_f32(const prost_hip_fused_desc* desc, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev,
                                      void* record, int use_kty, int use_kx_prev, int use_kty_prev, int cols_per_block, double* res_out4, void* workspace,
                                      int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_fused_iteration_rec_f64(const prost_hip_fused_desc* desc, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev,
                                      void* record, int use_kty, int use_kx_prev, int use_kty_prev, int cols_per_block, double* res_out4, void* workspace,
                                      int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
/* ... and of prost_hip_fused_iteration_mc (gradient2d with 3 / 4 channels) */
int prost_hip_fused_iteration_mc_rec_f32(const prost_hip_fused_desc* desc, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev,
                                         void* record, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4, void* workspace,
                                         int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_fused_iteration_mc_rec_f64(const prost_hip_fused_desc* desc, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev,
                                         void* record, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4, void* workspace,
                                         int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
/* ... and of prost_hip_fused_iteration_mc_x2 (two iterations per launch, 2-4 channels) */
int prost_hip_fused_iteration_mc_x2_rec_f32(const prost_hip_fused_desc* desc, float* x_out, float* y_out, const float* x, const float* y, void* record, int cols,
                                            double* res_out4, void* workspace, int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror,
                                            void* stream);
int prost_hip_fused_iteration_mc_x2_rec_f64(const prost_hip_fused_desc* desc, double* x_out, double* y_out, const double* x, const double* y, void* record, int cols,
                                            double* res_out4, void* workspace, int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror,
                                            void* stream);
/* ... and of the gradient3d kernels (round 5): prost_hip_fused_iteration3d (block_gradient3d.cu:25-150 inside one PerformIteration),
 * prost_hip_fused_iteration3d_pw (planes across the wavefronts of a workgroup; no residual sums) and prost_hip_fused_iteration3d_x2 (two
 * iterations per launch, both with the record's step sizes; residual sums of the second) */
int prost_hip_fused_iteration3d_rec_f32(const prost_hip_fused_desc* desc, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev, void* record,
                                        int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4, void* workspace, int apply_rule,
                                        unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_fused_iteration3d_rec_f64(const prost_hip_fused_desc* desc, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev, void* record,
                                        int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4, void* workspace, int apply_rule,
                                        unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_fused_iteration3d_pw_rec_f32(const prost_hip_fused_desc* desc, float* x_new, float* y_new, const float* x, const float* y, void* record, int use_kty,
                                           int use_kx_prev, int cols, int waves, void* stream);
int prost_hip_fused_iteration3d_pw_rec_f64(const prost_hip_fused_desc* desc, double* x_new, double* y_new, const double* x, const double* y, void* record, int use_kty,
                                           int use_kx_prev, int cols, int waves, void* stream);
int prost_hip_fused_iteration3d_x2_rec_f32(const prost_hip_fused_desc* desc, float* x_out, float* y_out, const float* x, const float* y, void* record, int cols,
                                           double* res_out4, void* workspace, int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_fused_iteration3d_x2_rec_f64(const prost_hip_fused_desc* desc, double* x_out, double* y_out, const double* x, const double* y, void* record, int cols,
                                           double* res_out4, void* workspace, int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_fused_iteration2_rec_f32(const prost_hip_fused_desc* desc, float* x_out, float* y_out, const float* x, const float* y, float* x_mid, float* y_mid,
                                       void* record, int cols_per_block, double* res_out4, void* workspace, int apply_rule, unsigned long long iteration,
                                       prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_fused_iteration2_rec_f64(const prost_hip_fused_desc* desc, double* x_out, double* y_out, const double* x, const double* y, double* x_mid, double* y_mid,
                                       void* record, int cols_per_block, double* res_out4, void* workspace, int apply_rule, unsigned long long iteration,
                                       prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_fused_iterationk_rec_f32(const prost_hip_fused_desc* desc, int k, float* x_out, float* y_out, const float* x, const float* y, void* record,
                                       int cols_per_block, double* res_out4, void* workspace, int apply_rule, unsigned long long iteration,
                                       prost_hip_pdhg_rule_state* mirror, void* stream);
int prost_hip_fused_iterationk_rec_f64(const prost_hip_fused_desc* desc, int k, double* x_out, double* y_out, const double* x, const double* y, void* record,
                                       int cols_per_block, double* res_out4, void* workspace, int apply_rule, unsigned long long iteration,
                                       prost_hip_pdhg_rule_state* mirror, void* stream);

/* The same iteration with the PLANES ACROSS THE WAVEFRONTS of a workgroup (kernels_fused_iter3d_pw.hip): `waves` - 1
 * consecutive planes per workgroup exchange x_new through LDS, one helper wavefront recomputes the plane above the
 * group (waves = 4 or 8; 0 = automatic).  No residual variant.  Needs L >= 2; otherwise the contract of
 * prost_hip_fused_iteration3d. */
int prost_hip_fused_iteration3d_pw_supported(const prost_hip_fused_desc* desc, int dtype /* 0 f32, 1 f64 */);
int prost_hip_fused_iteration3d_pw_f32(const prost_hip_fused_desc* desc, float* x_new, float* y_new, const float* x, const float* y, double tau, double sigma,
                                       double theta, int use_kty, int use_kx_prev, int cols, int waves, void* stream);
int prost_hip_fused_iteration3d_pw_f64(const prost_hip_fused_desc* desc, double* x_new, double* y_new, const double* x, const double* y, double tau, double sigma,
                                       double theta, int use_kty, int use_kx_prev, int cols, int waves, void* stream);

/* TWO iterations of a gradient2d problem with L = 2, 3 or 4 channels per launch (kernels_fused_iter_mc_x2.hip;
 * backend_pdhg.cu:317-370 twice): the channels on the wavefronts of a workgroup, each running the 4-stage column pipeline of
 * prost_hip_fused_iteration2, the squares of the dual arguments of both dual steps meeting in LDS for the norm over the
 * 2 L components of a pixel.  Reads x^k, y^k and b, writes x^(k+2), y^(k+2); the intermediate iterate is stored nowhere.
 * tau/sigma/theta: HOST arrays of 2.  Bit-identical to two prost_hip_fused_iteration_mc launches (resp. two
 * prost_hip_fused_iteration launches for L = 2).  Straight-line ROF / TV-L1 shapes (prox_g square or abs with scalar
 * a = 1, d = e = 0, b scalar or per pixel; prox_f* ind_leq0 with scalar a = 1, d = e = 0); any height (16 bytes of rows per lane
 * where it is a multiple of that, else one row). */
int prost_hip_fused_iteration_mc_x2_supported(const prost_hip_fused_desc* desc, int dtype /* 0 f32, 1 f64 */);
/* 1 iff the launch is also faster than two single launches (tiny images: one launch instead of two; large ones: half the HBM
 * traffic; in between -- about 384^2 to 700^2 RGB -- the single-iteration kernel is up to 9 % faster) */
int prost_hip_fused_iteration_mc_x2_profitable(const prost_hip_fused_desc* desc, int dtype);
/* (ABI 9) PROST_HIP_ARITH_* of the instance a launch with this description runs (FMAD: fp32, heights that are a multiple of 4, uniform Tau, ind_leq0 radius > 0) */
int prost_hip_fused_iteration_mc_x2_arith(const prost_hip_fused_desc* desc, int dtype);
int prost_hip_fused_iteration_mc_x2_chunk_cols(const prost_hip_fused_desc* desc, int dtype, int with_residuals);
/* res_out4 != NULL (needs `workspace`): also the four residual sums of the SECOND iteration, as prost_hip_fused_iteration_mc
 * writes them for that iteration (same terms, restricted to the owned columns res_x0 / res_x1; the summation order differs) */
int prost_hip_fused_iteration_mc_x2_f32(const prost_hip_fused_desc* desc, float* x_out, float* y_out, const float* x, const float* y, const double* tau,
                                        const double* sigma, const double* theta, int cols, double* res_out4, void* workspace, void* stream);
int prost_hip_fused_iteration_mc_x2_f64(const prost_hip_fused_desc* desc, double* x_out, double* y_out, const double* x, const double* y, const double* tau,
                                        const double* sigma, const double* theta, int cols, double* res_out4, void* workspace, void* stream);

/* TWO iterations of a gradient3d problem per launch (kernels_fused_iter3d_x2.hip; backend_pdhg.cu:317-370 twice): the
 * planes of a group run on the wavefronts of one workgroup, every wavefront runs the 4-stage column pipeline of
 * prost_hip_fused_iteration2 on its plane and the stages meet in LDS.  Reads x^k, y^k and b, writes x^(k+2), y^(k+2);
 * the intermediate iterate is stored nowhere, so callers pair only iterations whose intermediate state nobody observes.  tau/sigma/theta: HOST arrays of 2.  Bit-identical to two prost_hip_fused_iteration3d
 * launches (use_kty = use_kx_prev = 1).  fp32 and fp64, any height, straight-line ROF / TV-L1 shapes (prox_g square or abs with scalar a = 1, d = e = 0,
 * b scalar or per voxel; prox_f* ind_leq0 with scalar a = 1, d = e = 0): see _supported.
 * cols <= 0: automatic chunk length. */
int prost_hip_fused_iteration3d_x2_supported(const prost_hip_fused_desc* desc, int dtype /* 0 f32, 1 f64 */);
/* columns per chunk a launch with cols <= 0 uses (0 if unsupported): chosen so that the rounds of workgroups on the
 * device's compute units times the column steps of a workgroup is minimal */
/* (ABI 9) PROST_HIP_ARITH_* of the instance a launch with this description runs (FMAD: fp32, even heights, ind_leq0 radius > 0 and desc->arith asks for it) */
int prost_hip_fused_iteration3d_x2_arith(const prost_hip_fused_desc* desc, int dtype);
int prost_hip_fused_iteration3d_x2_chunk_cols(const prost_hip_fused_desc* desc, int dtype, int with_residuals);
/* res_out4 != NULL (needs `workspace`): also the four residual sums of the SECOND iteration, as
 * prost_hip_fused_iteration3d writes them for that iteration (same terms; the summation order differs) */
int prost_hip_fused_iteration3d_x2_f32(const prost_hip_fused_desc* desc, float* x_out, float* y_out, const float* x, const float* y, const double* tau,
                                       const double* sigma, const double* theta, int cols, double* res_out4, void* workspace, void* stream);
int prost_hip_fused_iteration3d_x2_f64(const prost_hip_fused_desc* desc, double* x_out, double* y_out, const double* x, const double* y, const double* tau,
                                       const double* sigma, const double* theta, int cols, double* res_out4, void* workspace, void* stream);

/* ONE kernel per iteration for gradient2d problems with L = 3 or 4 channels (kernels_fused_iter_mc.hip): the channels
 * run on the wavefronts of one workgroup and meet in LDS for the norm over the 2 L gradient components of a pixel
 * (sum_norm2(2 * nc, ...) of example_rof_primaldual.m).  Same contract as prost_hip_fused_iteration3d (incl. the
 * residual variant: y_prev, res_out4, workspace): outputs must not alias inputs, scalar coefficients except b of prox_g;
 * heights that are not a whole number of 16-byte row groups run a one-row-per-lane instance. */
int prost_hip_fused_iteration_mc_supported(const prost_hip_fused_desc* desc, int dtype /* 0 f32, 1 f64 */);
int prost_hip_fused_iteration_mc_f32(const prost_hip_fused_desc* desc, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev,
                                     double tau, double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4,
                                     void* workspace, void* stream);
int prost_hip_fused_iteration_mc_f64(const prost_hip_fused_desc* desc, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev,
                                     double tau, double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4,
                                     void* workspace, void* stream);

/* ONE kernel per iteration for gradient3d problems (kernels_fused_iter3d.hip): x_new = prox_g(x - tau T K^T y),
 * y_new = prox_f*(y + sigma S K(x_new + theta (x_new - x))) with x_new of plane l+1 recomputed in registers
 * (BackendPDHG::PerformIteration, backend_pdhg.cu:313-381, with BlockGradient3D, block_gradient3d.cu:25-150).
 * Outputs must not alias inputs.  Supported when ny is a multiple of the vector width (4 floats / 2 doubles), the
 * coefficients of prox_f* are scalars and at most b of prox_g is a per-voxel vector.  res_out4 != NULL (residual
 * iterations): y_prev = y^(k-1) is streamed too and res_out4 receives {primal diff^2, primal var^2, dual diff^2,
 * dual var^2} as prost_hip_fused_iteration does (workspace: prost_hip_reduce_workspace_bytes(); y_new must then not
 * alias y_prev).  cols = columns per wavefront chunk (0 = automatic). */
int prost_hip_fused_iteration3d_supported(const prost_hip_fused_desc* desc, int dtype /* 0 f32, 1 f64 */);
int prost_hip_fused_iteration3d_f32(const prost_hip_fused_desc* desc, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev,
                                    double tau, double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4,
                                    void* workspace, void* stream);
int prost_hip_fused_iteration3d_f64(const prost_hip_fused_desc* desc, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev,
                                    double tau, double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4,
                                    void* workspace, void* stream);
int prost_hip_fused_iteration2_f32(const prost_hip_fused_desc* desc, float* x_out, float* y_out, const float* x, const float* y,
                                   float* x_mid, float* y_mid, const double* tau, const double* sigma, const double* theta,
                                   int cols_per_block, double* res_out4, void* workspace, void* stream);
int prost_hip_fused_iteration2_f64(const prost_hip_fused_desc* desc, double* x_out, double* y_out, const double* x, const double* y,
                                   double* x_mid, double* y_mid, const double* tau, const double* sigma, const double* theta,
                                   int cols_per_block, double* res_out4, void* workspace, void* stream);

/* Device-side self test of the short correctly rounded forms the fused kernels rely on (csrc/device_math.hpp):
 * n / d through a refined fp64 reciprocal, sqrtf without range scaling on [2^-96, 2^126], the exact
 * (float)((double)x / D) of Function1DSquare, and float subtraction == double subtraction rounded to float.
 * Draws `n` pseudo-random cases (all exponents, subnormal / overflowing quotients, zeros) from `seed` and
 * writes the number of bit mismatches against the compiler's IEEE expansions to the DEVICE array
 * mismatches8 = {division, sqrt, exact division, subtraction, control, division with the product rounded through
 * fma(n, r, +0) against n / d + 0, min0(t) against t > 0 ? 0 : t, the fp64 quotient / square-root forms against `/` and sqrt
 * (ABI 6; unused before)}; `control` counts the cases where
 * the plain single-precision reciprocal product differs from n / d and must come out > 0. */
int prost_hip_selftest_math(unsigned long long* mismatches8, uint64_t n, uint64_t seed, void* stream);
/* Verification entry: comparison of two device vectors without a read-back (the 2048 x 2048 x 64 state is
 * 8 GB per vector set).  out2 (DEVICE double[2]) = {number of elements that differ in value (+0 == -0, NaN == NaN: what
 * numpy.array_equal(a, b, equal_nan=True) counts), sum |a_i - b_i| over them}; workspace: prost_hip_reduce_workspace_bytes().  The reference compares iterates on the host after
 * thrust::copy (backend_pdhg.cu:491-502); there is no device-side counterpart. */
int prost_hip_compare_f32(double* out2, const float* a, const float* b, size_t n, void* workspace, void* stream);
int prost_hip_compare_f64(double* out2, const double* a, const double* b, size_t n, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* ADMM / CGLS building blocks (src/backend/backend_admm.cu, include/prost/cgls.hpp)           */
/* ------------------------------------------------------------------------------------------ */
/* out[0] = sqrt(sum x_i^2), accumulated in double; out is a DEVICE double[2] (out[1] := 0)
 * (cgls.hpp:152-170 nrm2 via thrust; cublas{S,D}nrm2 at backend_admm.cu:40-50) */
int prost_hip_nrm2_f32(double* out, const float* x, size_t n, void* workspace, void* stream);
int prost_hip_nrm2_f64(double* out, const double* x, size_t n, void* workspace, void* stream);
/* y += alpha x  (cublas{S,D}axpy, cgls.hpp:141-150) */
int prost_hip_axpy_f32(float* y, const float* x, double alpha, size_t n, void* stream);
int prost_hip_axpy_f64(double* y, const double* x, double alpha, size_t n, void* stream);
/* Elementwise ADMM functors (backend_admm.cu:53-196), selected by `op`:
 *  PROST_ADMM_TEMP1   o = (alpha a + (1-alpha) b + c) / sqrt(d)                 temp1_functor
 *  PROST_ADMM_TEMP2   o = sqrt(c) (a + b)                                       temp2_functor
 *  PROST_ADMM_DIFF    o = a - b                                                 difference_functor
 *  PROST_ADMM_XPROJ   o = sqrt(b) (o + a)                                       x_proj_functor
 *  PROST_ADMM_XDUAL   o = a sqrt(c) - b                                         x_dual_functor
 *  PROST_ADMM_ZDUAL   o = a / sqrt(c) - b                                       z_dual_functor
 *  PROST_ADMM_GEMV1   o = sqrt(a) b                                             gemv_functor1
 *  PROST_ADMM_GEMV2   o = (beta / (alpha sqrt(a))) b     (alpha, beta scalars)  gemv_functor2
 *  PROST_ADMM_GEMV3   o = alpha sqrt(a) b                                       gemv_functor3
 *  PROST_ADMM_GETDUAL o = -alpha pow(d, beta) (a - b + c)   (alpha=rho,beta=expo) get_dual_functor
 *  PROST_ADMM_SCALE   o = alpha a                                               rescale :650-663
 *  PROST_ADMM_DIV     o = a / alpha                                             normest_divide problem.cu:411-421 */
enum { PROST_ADMM_TEMP1 = 0, PROST_ADMM_TEMP2, PROST_ADMM_DIFF, PROST_ADMM_XPROJ, PROST_ADMM_XDUAL, PROST_ADMM_ZDUAL,
       PROST_ADMM_GEMV1, PROST_ADMM_GEMV2, PROST_ADMM_GEMV3, PROST_ADMM_GETDUAL, PROST_ADMM_SCALE, PROST_ADMM_DIV };
int prost_hip_admm_elem_f32(int op, float* o, const float* a, const float* b, const float* c, const float* d, double alpha, double beta, size_t n, void* stream);
int prost_hip_admm_elem_f64(int op, double* o, const double* a, const double* b, const double* c, const double* d, double alpha, double beta, size_t n, void* stream);

/* Device-resident CGLS (include/prost/cgls.hpp:222-371 on the preconditioned operator
 * A = Sigma^(1/2) K Tau^(1/2) of GemvPrecondK, backend_admm.cu:199-272).  The CG scalars live in a device
 * record (`state`, prost_hip_cgls_state_bytes() bytes); each stage is one fused pass over the vectors
 * and the caller applies K / K^T between stages:
 *
 *   INIT_X ; INIT_R ; r += K t ; INIT_R2 ; s += K^T t ; INIT_S ;
 *   repeat maxit times:  q = K t ; STEP_Q ; STEP_XR ; s += K^T t ; STEP_S ; STEP_P
 *
 * After the stopping test of cgls.hpp:355-360 fires the STEP stages return without touching x, so the
 * host may launch all maxit rounds without waiting; STEP_S additionally stores `epoch` to the pinned
 * host word `host_done` (may be NULL) so that a host that is not running ahead can stop launching.
 * `tol` and `epoch` are read by INIT_X only (kept in the device record afterwards): the STEP launches of
 * one solver are argument-identical from solve to solve and can be replayed from a captured HIP graph.
 * b, r, q: m elements; x, p, s: n elements; t: max(m, n) scratch; sigma (m) / tau (n) are the
 * preconditioner diagonals (the stages take the square roots, as gemv_functor1-3 do). */
typedef struct prost_hip_cgls_desc {
  void* state;
  void* workspace;          /* prost_hip_cgls_workspace_bytes() */
  const void* b;
  void* x; void* p; void* q; void* r; void* s; void* t;
  const void* sigma; const void* tau;
  uint64_t m, n;
  double shift, tol;
  int* host_done;
  int epoch;
} prost_hip_cgls_desc;
typedef struct prost_hip_cgls_result_t {
  int iterations, converged, indefinite, flag;     /* flag 1: initial |s| < eps (cgls.hpp:283) */
  double norms, norms0, normx, xmax;
} prost_hip_cgls_result_t;
enum { PROST_CGLS_INIT_X = 0, PROST_CGLS_INIT_R, PROST_CGLS_INIT_R2, PROST_CGLS_INIT_S,
       PROST_CGLS_STEP_Q, PROST_CGLS_STEP_XR, PROST_CGLS_STEP_S, PROST_CGLS_STEP_P };
size_t prost_hip_cgls_state_bytes(void);
size_t prost_hip_cgls_workspace_bytes(void);
int prost_hip_cgls_stage_f32(int stage, const prost_hip_cgls_desc* d, void* stream);
int prost_hip_cgls_stage_f64(int stage, const prost_hip_cgls_desc* d, void* stream);
/* blocking read-back of the scalar record (synchronises `stream`) */
int prost_hip_cgls_result(const void* state, prost_hip_cgls_result_t* out, void* stream);

/* A CG round in FOUR launches for operators made of CSR and gradient blocks (the shapes of block_sparse.cu:146-211 and
 * block_gradient2d.cu:26-139 / block_gradient3d.cu:25-150): the operator of GemvPrecondK is applied by one kernel per
 * direction -- the thread that owns an output element evaluates its CSR row / stencil row for every block that covers it, in
 * block order, as LinearOperator::Eval / EvalAdjoint accumulate them (linearoperator.cu:135-170) -- with the stage that follows
 * the product as its epilogue, and the two scalar kernels of the round above are folded into the kernels that consume their
 * results (every workgroup forms alpha / beta / the stopping test of cgls.hpp:297-360 from the partial sums itself):
 *
 *   round j:  q = sqrt(Sigma) K t, |q|^2 ; alpha, x += alpha p, r -= alpha q, s = -shift x / sqrt(Tau), t = sqrt(Sigma) r, |x|^2 ;
 *             s = sqrt(Tau) (s + K^T t), |s|^2 ; beta, stopping test, p = beta p + s, t = sqrt(Tau) p, |p|^2
 *
 * `state` is an ARRAY of records here (prost_hip_cgls_state_bytes() each): the INIT stages of prost_hip_cgls_stage write
 * record 0, round j reads record j and writes record j + 1 (rounds after the stopping test fired only hand the record on), so
 * a solve of at most R rounds needs R + 1 records and its result is record R (prost_hip_cgls_result_at).  Same per-element
 * expressions and roundings as the staged round; the sums are order-independent (reduce.hpp: exact to 2^-100, rounded once), so
 * every CG scalar equals the staged round's.
 * CSR blocks: one thread per row, sequential sum -- what prost_hip_csr_spmv does for rows of up to 6 entries on average.
 * Gradient blocks: planar layout (label_first = false). */
enum { PROST_OP_CSR = 1, PROST_OP_GRAD2D = 2, PROST_OP_GRAD3D = 3 };
typedef struct prost_hip_op_block {
  int kind;
  uint64_t row, col, nrows, ncols;          /* position and size inside the operator (Block::row() ...) */
  uint64_t nx, ny, L;                       /* gradient blocks */
  const void* val; const int32_t* ptr; const int32_t* ind;          /* CSR of K   (T values, nrows + 1 row starts) */
  const void* val_t; const int32_t* ptr_t; const int32_t* ind_t;    /* CSR of K^T (ncols + 1 row starts) */
  /* ABI 7: a sparse block whose product runs from ROW PATTERNS (prost_hip_pattern_spmv: one 16-bit pattern number per row + a table of
   * (column - row, value) sequences) instead of CSR arrays, for K and / or K^T; ids / ids_t NULL: the CSR arrays above */
  const uint16_t* ids; const int32_t* pptr; const int32_t* rel; const void* pval;
  const uint16_t* ids_t; const int32_t* pptr_t; const int32_t* rel_t; const void* pval_t;
  /* (ABI 8) ANCHORED tables: the offsets of a pattern count from anchor[row] (the row's first column) instead of from the row number
   * (prost_hip_pattern_spmv_anchored); NULL: offsets relative to the row */
  const int32_t* anchor; const int32_t* anchor_t;
} prost_hip_op_block;
#define PROST_HIP_OP_MAX_BLOCKS 4
typedef struct prost_hip_fused_op {
  int nblocks;
  prost_hip_op_block block[PROST_HIP_OP_MAX_BLOCKS];
} prost_hip_fused_op;
/* 1 if the rounds take this operator of m rows and n columns (host-only check, no launch) */
int prost_hip_fused_op_supported(const prost_hip_fused_op* op, uint64_t m, uint64_t n);
int prost_hip_cgls_round_f32(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, int round, void* stream);
int prost_hip_cgls_round_f64(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, int round, void* stream);
/* The same round with per-kernel timing: kernel k of the round (0: the forward operator stage, 1: STEP_XR2, 2: the adjoint operator
 * stage, 3: STEP_P2) stamps ev8[2 k] / ev8[2 k + 1] (events of prost_hip_event_create; a NULL pair skips that kernel) with its
 * own begin / end, as prost_hip_next_launch_events does: no marker packets between the launches. */
int prost_hip_cgls_round_timed_f32(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, int round, void* const* ev8, void* stream);
int prost_hip_cgls_round_timed_f64(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, int round, void* const* ev8, void* stream);
/* A CG round in TWO launches for operators K = [D ; gradient2d(nx, ny, L)] (or [gradient2d ; D], or the gradient alone): D is an
 * (nx ny) x (L nx ny) sparse block whose row i holds exactly L entries, at columns i + c nx ny, c = 0 .. L-1 -- a matrix that couples
 * the channels of ONE pixel, e.g. the warp matrix [diag(Ix) diag(Iy)] of a TV-L1 flow model.  A thread owns 4 consecutive pixels of an
 * image column (2 in fp64) with all their rows and columns of K; what a stage needs from neighbouring pixels it recomputes from their
 * operands, so the vector updates of cgls.hpp:311-351 fold into the operator stages (block_sparse.cu:146-211, block_gradient2d.cu:26-139,
 * backend_admm.cu:199-272):
 *   launch A: beta, stopping test (of the previous round) ; p = beta p + s ; q = sqrt(Sigma) K sqrt(Tau) p ; |p|^2, |q|^2
 *   launch B: alpha ; x += alpha p ; r -= alpha q ; s = sqrt(Tau) (-shift x / sqrt(Tau) + K^T sqrt(Sigma) r) ; |x|^2, |s|^2
 * Same records as prost_hip_cgls_round (record 0 from prost_hip_cgls_init_fused; launch A of round j writes record j); after the
 * last queued round `last`, prost_hip_cgls_pixel_close writes record last + 1 (the evaluation launch A of the next round would
 * make), so prost_hip_cgls_result_at(state, rounds queued) reads what the other paths leave there.  p and r alternate between the
 * descriptor's buffers and p_alt / r_alt (n and m elements); d->t is not used.  Every vector is bit-identical to the four-launch
 * round's: same per-element expressions in the same order, order-independent sums. */
typedef struct prost_hip_pixel_op {
  uint64_t nx, ny;
  int L;                    /* channels, 1 .. 3 */
  int has_d;                /* 0: K is the gradient block alone (m = 2 L nx ny) */
  int d_first;              /* D stands BEFORE the gradient block in the operator's block list: K^T t is accumulated block by block in list
                             * order (LinearOperator::EvalAdjoint, linearoperator.cu:152-170), whatever the row order is */
  uint64_t d_row, g_row;    /* first row of D (nx ny rows) and of the gradient block (2 L nx ny rows): {0, nx ny} or {2 L nx ny, 0} */
  const void* w;            /* D's values, row-major: w[i L + c] = D(i, i + c nx ny) -- the value array of D's CSR form (T) */
  void* p_alt; void* r_alt; /* second buffers for p (n elements) and r (m elements) */
  double sigma_grad;        /* Sigma on the gradient rows: ONE value (the caller checks that d->sigma is constant there; a gradient
                             * block's row sums are, block_gradient2d.cu:154-158) -- d->sigma is read on D's rows only */
  /* ABI 10: D as ANY CSR block of nx ny rows over the L nx ny primal entries (block_sparse.cu:146-211: a warp matrix whose rows gather at
   * displaced pixels).  d_csr = 1 (with has_d = 1): the rows of D come from d_val / d_ptr / d_ind, its columns from the CSR arrays of D^T
   * (values of the solve's type, 32-bit indices, device memory; w is not read).  The thread that owns a pixel owns D's row of that pixel
   * and the L primal entries of that pixel; the operand of every entry -- sqrt(Tau) p, sqrt(Sigma) r of ANOTHER pixel, updated in this
   * very launch -- is recomputed from that pixel's stored operands.  Same expressions in the same order as the four-launch round. */
  int d_csr;
  const void* d_val; const int32_t* d_ptr; const int32_t* d_ind;
  const void* dt_val; const int32_t* dt_ptr; const int32_t* dt_ind;
} prost_hip_pixel_op;
/* 1 if the two-launch rounds take this operator (host-only check); dtype 0: fp32 (ny % 4 == 0), 1: fp64 (ny % 2 == 0) */
int prost_hip_pixel_op_supported(const prost_hip_pixel_op* op, uint64_t m, uint64_t n, int dtype);
int prost_hip_cgls_pixel_round_f32(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int round, void* stream);
int prost_hip_cgls_pixel_round_f64(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int round, void* stream);
/* with per-kernel timing: launch A stamps ev4[0] / ev4[1], launch B ev4[2] / ev4[3] (as prost_hip_cgls_round_timed) */
int prost_hip_cgls_pixel_round_timed_f32(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int round, void* const* ev4, void* stream);
int prost_hip_cgls_pixel_round_timed_f64(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int round, void* const* ev4, void* stream);
int prost_hip_cgls_pixel_close_f32(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int last_round, void* stream);
int prost_hip_cgls_pixel_close_f64(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int last_round, void* stream);
/* blocking read-back of record `index` of a record array */
int prost_hip_cgls_result_at(const void* state, int index, prost_hip_cgls_result_t* out, void* stream);
/* The start of a solve the same way: INIT_X ; [INIT_R ; r += K t ; INIT_R2] ; [s += K^T (sqrt(Sigma) r) ; INIT_S], each bracket
 * one kernel; sqrt(Sigma) r passes through q (which the first round overwrites).  Writes record 0. */
int prost_hip_cgls_init_fused_f32(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, void* stream);
int prost_hip_cgls_init_fused_f64(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, void* stream);

/* Fused passes of the ADMM outer iteration (BackendADMM::PerformIteration, backend_admm.cu:355-665): each
 * stage applies, per element, the reference functors that touch that element (same expressions, same
 * order) and replaces the device-to-device copies between them.  The caller applies K / K^T in between:
 *
 *   PRE_X ; PRE_Z ; z_dual += K temp3 ; PRE_Z2 ; [CGLS on b = z_dual, x = x_proj] ;
 *   POST_X ; z_proj = K x_proj ; POST_XZ ; prox_g(temp1) -> x_half ; prox_f(temp2) -> z_half ;
 *   residuals (:535-616):  kx = K x_half ; RES_Z (overwrites kx with the dual variable y) ;
 *                          kty = K^T kx ; RES_X -> out4 = {primal residual, primal variable norm,
 *                          dual residual, dual variable norm} (doubles; out4 may be pinned host memory)
 * x_*: n elements, z_*: m elements, temp1: n, temp2: m, temp3: max(m, n), kx: m, kty: n. */
typedef struct prost_hip_admm_desc {
  void* workspace;          /* prost_hip_cgls_workspace_bytes() */
  void* x_half; void* x_proj; void* x_dual;
  void* z_half; void* z_proj; void* z_dual;
  void* temp1; void* temp2; void* temp3;
  void* kx; void* kty;
  const void* sigma; const void* tau;
  uint64_t m, n;
  double alpha;             /* over-relaxation */
  double rho;
  double* out4;
} prost_hip_admm_desc;
enum { PROST_ADMM_STAGE_PRE_X = 0, PROST_ADMM_STAGE_PRE_Z, PROST_ADMM_STAGE_PRE_Z2, PROST_ADMM_STAGE_POST_X,
       PROST_ADMM_STAGE_POST_XZ, PROST_ADMM_STAGE_RES_Z, PROST_ADMM_STAGE_RES_X };
int prost_hip_admm_stage_f32(int stage, const prost_hip_admm_desc* d, void* stream);
int prost_hip_admm_stage_f64(int stage, const prost_hip_admm_desc* d, void* stream);
/* The ADMM outer iteration with the operator inside the stages (prost_hip_admm_stage: same expressions, the caller applies
 * K between the stages there).  Each bracket is one kernel:
 *   PRE : PRE_X ; [PRE_Z ; z_dual += K temp3 ; PRE_Z2]
 *   POST: [POST_X ; the n half of POST_XZ] ; [z_proj = K x_proj ; the m half of POST_XZ]
 *   RES : [kx = K x_half ; RES_Z] ; [K^T kx (not stored) ; RES_X] ; fold -> out4 */
enum { PROST_ADMM_FUSED_PRE = 0, PROST_ADMM_FUSED_POST, PROST_ADMM_FUSED_RES };
int prost_hip_admm_fused_stage_f32(int stage, const prost_hip_admm_desc* d, const prost_hip_fused_op* op, void* stream);
int prost_hip_admm_fused_stage_f64(int stage, const prost_hip_admm_desc* d, const prost_hip_fused_op* op, void* stream);

/* Fused passes of the operator-norm power iteration (Problem::normest, problem.cu:429-500), one round:
 *   NORMEST_A: x_temp = sqrt(tau) (x / norm_x)        (norm_x = 0: no divide -- first round)
 *   ax = K x_temp ; NORMEST_B: a = sqrt(sigma) ax, out[0] = |a|, ax = sqrt(sigma) a
 *   x_temp = K^T ax ; NORMEST_C: x = sqrt(tau) x_temp, out[1] = |x|
 * x, x_temp: n elements; ax: m; out: two doubles (device or pinned host); workspace: prost_hip_cgls_workspace_bytes(). */
typedef struct prost_hip_normest_desc {
  void* workspace;
  void* x; void* x_temp; void* ax;
  const void* sigma; const void* tau;
  uint64_t m, n;
  double norm_x;
  double* out;
  const double* norm_x_from;   /* NORMEST_A: if not NULL the divisor is read from here (device-visible memory, e.g. the out[1] a
                                * previous round wrote) instead of norm_x -- rounds can be queued without a host round trip */
} prost_hip_normest_desc;
enum { PROST_NORMEST_A = 0, PROST_NORMEST_B, PROST_NORMEST_C };
int prost_hip_normest_stage_f32(int stage, const prost_hip_normest_desc* d, void* stream);
int prost_hip_normest_stage_f64(int stage, const prost_hip_normest_desc* d, void* stream);

/* The whole round in ONE kernel when the operator is a single gradient2d / gradient3d block (not label_first) under constant
 * preconditioners tau, sigma (kernels_normest_grad.hip): x_out = sqrt(tau) K^T sigma K sqrt(tau) (x_in / norm), out[0] = |a|,
 * out[1] = |x_out| as the three stages above report them.  x_out is bit-identical to the staged round (every intermediate
 * value is formed by the same expression); the norms sum the same terms in another order.  x_in != x_out (n = nx ny L
 * elements each); norm_x_from: NULL or 0.0 there = first round, no divide; workspace: prost_hip_cgls_workspace_bytes(). */
typedef struct prost_hip_normest_grad_desc {
  int is3d;
  uint64_t nx, ny, L;
  const void* x_in; void* x_out;
  double tau, sigma;
  const double* norm_x_from;
  double* out;
  void* workspace;
} prost_hip_normest_grad_desc;
int prost_hip_normest_grad_round_f32(const prost_hip_normest_grad_desc* d, void* stream);
int prost_hip_normest_grad_round_f64(const prost_hip_normest_grad_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* multi-GPU: global stopping criterion (no counterpart in the reference)        */
/* ------------------------------------------------------------------------------------------ */
/* RCCL communicator over the ranks of one node; unique_id is the 128-byte ncclUniqueId made
 * by rank 0 (prost_hip_comm_unique_id) and broadcast by the launcher. */
int prost_hip_comm_unique_id(void* id128);
int prost_hip_comm_create(void** comm, const void* id128, int rank, int world_size);
int prost_hip_comm_destroy(void* comm);
/* Host-callback transport: a communicator whose all-reduce is `fn(user, values, count)` -- sum `values` (pinned host
 * memory, in place) over the ranks, e.g. with gloo or MPI on the host.  prost_hip_allreduce_sum_f64 on it enqueues a
 * D2H copy, the function (hipLaunchHostFunc: it runs on a runtime thread and must not call HIP) and the H2D copy on the
 * given stream, i.e. it keeps the enqueue-and-return contract of the RCCL call, so the callers' stream / event logic is
 * the same on both.  For running several ranks on ONE GPU (RCCL refuses duplicate devices): tests of the N > 1 host
 * logic on a single-GPU box.  The reference has no multi-process path at all (prost.cpp:299-303: set_gpu). */
typedef void (*prost_hip_host_allreduce_fn)(void* user, double* values, size_t count);
int prost_hip_comm_create_host(void** comm, prost_hip_host_allreduce_fn fn, void* user);
/* Point-to-point on the host-callback transport: `p2p(user, nops, is_send, peers, bufs, bytes)` performs ALL transfers of one
 * group -- operation i sends (is_send[i] != 0) or receives bytes[i] bytes of pinned host memory bufs[i] to / from rank
 * peers[i] -- and returns when every one of them is complete (e.g. gloo isend / irecv + wait).  prost_hip_comm_send / recv
 * on such a communicator collect the operations of a group; group_end (or the call itself outside a group) enqueues the D2H
 * copies of what is sent, the function (hipLaunchHostFunc: runtime thread, no HIP calls) and the H2D copies of what is
 * received on the stream -- the enqueue-and-return contract of an RCCL group.  One communicator and one stream per group;
 * consecutive groups of a communicator must be ordered by their stream (they share a staging area).  world_size is what
 * prost_hip_comm_count reports for the communicator.  p2p may be NULL (all-reduce only). */
typedef void (*prost_hip_host_p2p_fn)(void* user, int nops, const int* is_send, const int* peers, void* const* bufs, const size_t* bytes);
int prost_hip_comm_host_configure(void* comm, int world_size, prost_hip_host_p2p_fn p2p, void* p2p_user);
/* number of ranks of the communicator (ncclCommCount; the configured world size of a host-callback communicator) */
int prost_hip_comm_count(void* comm, int* nranks);
/* 1 for a host-callback communicator, 0 for RCCL */
int prost_hip_comm_is_host(void* comm);
/* in-place sum all-reduce of `count` DEVICE doubles (the 4 residual sums) on `stream` */
int prost_hip_allreduce_sum_f64(void* comm, double* buf, size_t count, void* stream);
/* point-to-point transfers of `bytes` bytes of device memory with rank `peer` of the communicator (RCCL
 * ncclSend / ncclRecv over xGMI; the p2p function of a host-callback communicator).  Issue the sends and receives of one exchange step between group_start
 * and group_end: they then complete as ONE group, so both neighbours can be served without deadlock.
 * Used for the halo columns of column-sharded images. */
int prost_hip_comm_group_start(void);
int prost_hip_comm_group_end(void);
int prost_hip_comm_send(void* comm, const void* buf, size_t bytes, int peer, void* stream);
int prost_hip_comm_recv(void* comm, void* buf, size_t bytes, int peer, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PROST_HIP_H_ */
