// kernels_prox.hip -- generic proximal-operator kernels for gfx950.
//
// One separable element per lane (prox_elem_operation.inl:59-94).  For the planar layout
// (interleaved = false) component i of element tx lives at tx + count*i: lanes of a wave read
// 64 consecutive values per component -> fully coalesced.  The function id is wave-uniform, so
// the 14-way dispatch is one scalar branch per wave (no per-function template explosion: the
// reference instantiates 176 kernels, prox_elem_operation.cu:44-256; this file compiles 4).
#include "elementwise.hpp"
#include "fused_op.hpp"
#include "reduce.hpp"

namespace prost_hip {

template <class T>
struct Coeffs {
  const T* ptr[7];
  T val[7];
};

// Where the prox argument comes from.  ARG = 0: the `arg` vector.  ARG = 1 / 2: the PDHG prox arguments are formed
// on the fly from the iterate and the operator product, with the expressions of primal_proxarg_functor /
// dual_proxarg_functor (backend_pdhg.cu:38-70) -- the separate argument pass (write + re-read of one vector per
// prox) disappears, the values are bit-identical.
template <class T>
struct ArgSrc {
  const T* v0; const T* v1; const T* v2; const T* v3;      // ARG 0: v0 = arg.  1: x, T, K^T y.  2: y, Sigma, K x, K x_prev
  T s0, s1;                                                //                   1: tau          2: sigma, theta
  // ARG 1 / 2 inside a batch of iterations with device-resident step sizes (prost_hip_use_step_record): s0, s1 and the prox's own
  // step size are read from the record on the device, and the launch is a no-op once the record's stop flag is raised
  const PdhgRecord<T>* rec;
};
// ARG = 3 / 4 (round 5): the PDHG prox arguments of ARG 1 / 2 with the OPERATOR PRODUCT formed on the fly as well (fused_op.hpp): K^T y
// (ARG 3) resp. K x and K x_prev (ARG 4) are evaluated, for the elements a lane owns, from the operator's blocks -- row patterns, CSR
// rows, gradient stencils -- in the order LinearOperator::Eval / EvalAdjoint accumulate them, so the argument has the bits of the
// separate passes and K x, K^T y are never written (the reference: fill + K + fill + K^T, two full vectors through memory per
// iteration; backend_pdhg.cu:334-347, :370-380).  The same launch can add up the terms of the residuals
// (primal_residual_transform / dual_residual_transform, backend_pdhg.cu:73-120) of its elements: every operand is in registers.
template <class T> struct OpSrc {
  const FusedOpDev* opp;       // the block table, in device memory (fused_op.hpp, device_op)
  size_t base;                 // index of the prox's first element in the whole variable: a column of K (ARG 3) / a row (ARG 4)
  const T* w0; const T* w1;    // ARG 3: w0 = y (whole dual vector) ; ARG 4: w0 = x, w1 = x_prev (whole primal vectors)
  T* kty_out;                  // ARG 3: K^T y of the range is stored here (prox range; the NEXT iteration's dual residual reads it); may be null
  int use0, use1;              // 0: that product counts as the zero vector (iterations 0 / 1 of the reference, backend_pdhg.cu:213-216)
  double* res_ws;              // one (a.hi, a.lo, b.hi, b.lo) slot per workgroup, from slot res_slot on (reduce.hpp); null: no residual sums
  unsigned res_slot;
};
struct NoOpSrc {};
template <class T, int ARG> struct OpSrcOf { typedef NoOpSrc type; };
template <class T> struct OpSrcOf<T, 3> { typedef OpSrc<T> type; };
template <class T> struct OpSrcOf<T, 4> { typedef OpSrc<T> type; };
// ARG = 5 / 6: ARG 1 / 2 (the products K^T y resp. K x, K x_prev read from memory) with the residual sums of ARG 3 / 4 -- the launch that has
// x_prev, x, T, K^T y (resp. y_prev, y, Sigma, K x, K x_prev) in registers adds up the terms of dual_residual_transform / primal_residual_
// transform itself; only K^T y_prev (ARG 5: a.v3) is read in addition.  The separate reduction re-read eight vectors per residual iteration.
template <class T> struct OpSrcOf<T, 5> { typedef OpSrc<T> type; };
template <class T> struct OpSrcOf<T, 6> { typedef OpSrc<T> type; };
constexpr bool arg_is_primal(int ARG) { return ARG == 1 || ARG == 3 || ARG == 5; }
// the operands an argument was formed from (ARG 3 / 4), kept for the residual terms
template <class T, int VEC> struct ArgParts { T p0[VEC], p1[VEC], p2[VEC], p3[VEC]; };

template <class T, int ARG>
__device__ __forceinline__ bool steps_from_record(ArgSrc<T>& a, T& tau_scal) {
  if (ARG == 0 || !a.rec) return true;
  if (a.rec->stop) return false;
  a.s0 = arg_is_primal(ARG) ? a.rec->p.tau : a.rec->p.sigma;
  a.s1 = a.rec->p.theta;
  tau_scal = a.s0;
  return true;
}
template <class T, int ARG>
__device__ __forceinline__ T arg_formula(const ArgSrc<T>& a, T p0, T p1, T p2, T p3) {
  if (arg_is_primal(ARG)) return p0 - a.s0 * p1 * p2;
  return p0 + a.s0 * p1 * ((1 + a.s1) * p2 - a.s1 * p3);
}
// ARG 3 / 4: the argument of VEC consecutive elements from `off` on, its operands kept in `P`
template <class T, int VEC, int ARG>
__device__ __forceinline__ void load_arg_op(const ArgSrc<T>& a, const OpSrc<T>& os, size_t off, T (&out)[VEC], ArgParts<T, VEC>& P, bool store = true) {
  ldv<T, VEC>(a.v0 + off, P.p0); ldv<T, VEC>(a.v1 + off, P.p1);
#pragma unroll
  for (int j = 0; j < VEC; j++) { P.p2[j] = 0; P.p3[j] = 0; }
  const size_t g = os.base + off;
  if (ARG == 5) {                                                                   // K^T y from memory; K^T y_prev only for the residual terms
    ldv<T, VEC>(a.v2 + off, P.p2);
    if (os.res_ws && a.v3) ldv<T, VEC>(a.v3 + off, P.p3);
  } else if (ARG == 6) {                                                            // K x, K x_prev from memory
    ldv<T, VEC>(a.v2 + off, P.p2); ldv<T, VEC>(a.v3 + off, P.p3);
  } else if (ARG == 3) {
    if (os.res_ws && a.v3) ldv<T, VEC>(a.v3 + off, P.p3);                          // K^T y_prev, stored by the previous iteration's launch
    if (os.use0) op_adj_cols<T, VEC>(*as_constant(os.opp), g, g, os.w0, P.p2, false);            // 0 + K^T y, block after block
    if (store && os.kty_out) stv<T, VEC>(os.kty_out + off, P.p2);
  } else {
    if (os.use0 && os.use1) {                                                                  // K x and K x_prev in one walk over the blocks
      const T* tt[2] = {os.w0, os.w1};
      T kk[2][VEC];
      op_fwd_rows_n<T, VEC, 2>(*as_constant(os.opp), g, g, tt, kk, false);
#pragma unroll
      for (int j = 0; j < VEC; j++) { P.p2[j] = kk[0][j]; P.p3[j] = kk[1][j]; }
    } else {
      if (os.use0) op_fwd_rows<T, VEC>(*as_constant(os.opp), g, g, os.w0, P.p2, false);          // K x
      if (os.use1) op_fwd_rows<T, VEC>(*as_constant(os.opp), g, g, os.w1, P.p3, false);          // K x_prev
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; j++) out[j] = arg_formula<T, ARG>(a, P.p0[j], P.p1[j], P.p2[j], !arg_is_primal(ARG) ? P.p3[j] : (T)0);
}
// the residual terms of VEC elements whose prox result is `res` (ResidualPrimalF / ResidualDualF of kernels_pdhg.hip, term by term)
template <class T, int VEC, int ARG>
__device__ __forceinline__ void residual_terms(const ArgSrc<T>& a, const ArgParts<T, VEC>& P, const T (&res)[VEC], dd_t& sa, dd_t& sb) {
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    const T d = P.p1[j];
    if (!arg_is_primal(ARG)) {                         // in = y_prev, y, Sigma, K x_prev, K x
      const T z_hat = (P.p0[j] - res[j]) / (a.s0 * t_sqrt(d)) + t_sqrt(d) * ((1 + a.s1) * P.p2[j] - a.s1 * P.p3[j]);
      const T diff = z_hat - t_sqrt(d) * P.p2[j];
      dd_acc(sa, (double)(diff * diff)); dd_acc(sb, (double)(z_hat * z_hat));
    } else {                                           // in = x_prev, x, T, K^T y_prev, K^T y
      const T w_hat = (P.p0[j] - res[j]) / (a.s0 * t_sqrt(d)) - t_sqrt(d) * P.p3[j];
      const T diff = w_hat + t_sqrt(d) * P.p2[j];
      dd_acc(sa, (double)(diff * diff)); dd_acc(sb, (double)(w_hat * w_hat));
    }
  }
}
template <class T, int VEC, int ARG>
__device__ __forceinline__ void load_arg(const ArgSrc<T>& a, size_t off, T (&out)[VEC]) {
  if (ARG == 0) { ldv<T, VEC>(a.v0 + off, out); return; }
  T p0[VEC], p1[VEC], p2[VEC], p3[VEC];
  ldv<T, VEC>(a.v0 + off, p0); ldv<T, VEC>(a.v1 + off, p1); ldv<T, VEC>(a.v2 + off, p2);
  if (ARG == 2) ldv<T, VEC>(a.v3 + off, p3);
#pragma unroll
  for (int j = 0; j < VEC; j++) out[j] = arg_formula<T, ARG>(a, p0[j], p1[j], p2[j], ARG == 2 ? p3[j] : (T)0);
}
template <class T, int ARG>
__device__ __forceinline__ T load_arg1(const ArgSrc<T>& a, size_t off) {
  if (ARG == 0) return a.v0[off];
  return arg_formula<T, ARG>(a, a.v0[off], a.v1[off], a.v2[off], ARG == 2 ? a.v3[off] : (T)0);
}

// MOREAU: the prox of the CONJUGATE in the same pass (ProxMoreau::EvalLocal, prox_moreau.cu:98-134, around an
// elem operation): arg is pre-scaled per element (MoreauPrescale :29-43), the elem operation runs with the inverted
// step, and the result is post-scaled (MoreauPostscale :45-61) -- 3 values per element through HBM instead of 9.
// `invert_tau` is then the flag the Moreau wrapper itself was called with.
template <class T> __device__ __forceinline__ T moreau_pre(T a, T tau, T td, bool inv) { return inv ? a * (tau * td) : a / (tau * td); }
template <class T> __device__ __forceinline__ T moreau_post(T a, T r, T tau, T td, bool inv) { return inv ? a - r / (tau * td) : a - tau * td * r; }

template <class T, int OP, bool MOREAU, int ARG>
__global__ void __launch_bounds__(kBlock) prox_elem_kernel(T* __restrict__ res, ArgSrc<T> arg,
                                                           const T* __restrict__ tau_diag, T tau_scal, bool invert_tau,
                                                           size_t count, size_t dim, bool interleaved, int fn, Coeffs<T> cf) {
  if (!steps_from_record<T, ARG>(arg, tau_scal)) return;
  const bool inner_inv = MOREAU ? !invert_tau : invert_tau;      // step flag the elem operation sees
  for (size_t tx = (size_t)blockIdx.x * kBlock + threadIdx.x; tx < count; tx += (size_t)gridDim.x * kBlock) {
    T c[7];
#pragma unroll
    for (int i = 0; i < 7; i++) c[i] = cf.ptr[i] ? cf.ptr[i][tx] : cf.val[i];
    if (OP == PROST_OP_1D) {
      // Vector index with dim = 1: both layouts give tx (vector.hpp:44-48)
      const T td = tau_diag[tx], a = load_arg1<T, ARG>(arg, tx);
      const T tau = elem_tau<T>(tau_scal, td, inner_inv);
      const T r = elem_1d<T>(fn, MOREAU ? moreau_pre<T>(a, tau_scal, td, invert_tau) : a, tau, c);
      res[tx] = MOREAU ? moreau_post<T>(a, r, tau_scal, td, invert_tau) : r;
    } else {
      // ElemOperationNorm2 (elem_operation_norm2.hpp:40-88)
      const size_t base = interleaved ? tx * dim : tx;
      const size_t stride = interleaved ? 1 : count;
      T norm = 0;
      for (size_t i = 0; i < dim; i++) {
        T v = load_arg1<T, ARG>(arg, base + i * stride);
        if (MOREAU) v = moreau_pre<T>(v, tau_scal, tau_diag[base + i * stride], invert_tau);
        norm += v * v;
      }
      T pr = 0;
      if (norm > 0) {
        norm = t_sqrt(norm);
        const T tau = elem_tau<T>(tau_scal, tau_diag[base], inner_inv);   // tau_diag[0] only (:61)
        pr = scaled_prox<T>(fn, norm, tau, c);
      }
      for (size_t i = 0; i < dim; i++) {
        const T a = load_arg1<T, ARG>(arg, base + i * stride);
        const T td = MOREAU ? tau_diag[base + i * stride] : (T)0;
        const T v = MOREAU ? moreau_pre<T>(a, tau_scal, td, invert_tau) : a;
        const T r = norm > 0 ? pr * v / norm : (T)0;
        res[base + i * stride] = MOREAU ? moreau_post<T>(a, r, tau_scal, td, invert_tau) : r;
      }
    }
  }
}

// Vectorised form for the planar layout (and the 1-D operation): a lane owns 16 bytes of
// consecutive elements per component (float4 / double2), so every access of a wave is one 1-KiB
// transaction and the per-element coefficient vectors are read with the same width.  DIM > 0 keeps
// the components in registers; DIM == 0 (any dimension) makes a second pass over arg (L2 hits).
template <class T, int OP, int DIM, bool MOREAU, int ARG>
__global__ void __launch_bounds__(kBlock, ARG >= 3 && sizeof(T) == 4 ? (DIM <= 1 ? 4 : DIM == 2 ? 3 : 1) : 1) prox_elem_vec_kernel(T* __restrict__ res, ArgSrc<T> arg,
                                                               const T* __restrict__ tau_diag, T tau_scal, bool invert_tau,
                                                               size_t count, size_t dim, int fn, Coeffs<T> cf, bool e_zero, bool a_one,
                                                               typename OpSrcOf<T, ARG>::type os) {
  if (!steps_from_record<T, ARG>(arg, tau_scal)) return;
  constexpr int VEC = VecOf<T>::N;
  constexpr int D = DIM > 0 ? DIM : 1;
  constexpr bool OPA = ARG >= 3;                       // operator product on the fly; residual sums on request
  dd_t r_a{0.0, 0.0}, r_b{0.0, 0.0};
  bool with_res = false;
  if constexpr (OPA) with_res = os.res_ws != nullptr;
  for (size_t t0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * VEC; t0 < count; t0 += (size_t)gridDim.x * kBlock * VEC) {
    T c[7][VEC], td[VEC];
#pragma unroll
    for (int k = 0; k < 7; k++) {
      if (cf.ptr[k]) ldv<T, VEC>(cf.ptr[k] + t0, c[k]);
      else {
#pragma unroll
        for (int j = 0; j < VEC; j++) c[k][j] = cf.val[k];
      }
    }
    ldv<T, VEC>(tau_diag + t0, td);
    const bool inner_inv = MOREAU ? !invert_tau : invert_tau;
    if (OP == PROST_OP_1D) {
      T a[VEC], out[VEC];
      ArgParts<T, OPA ? VEC : 1> parts;
      if constexpr (OPA) load_arg_op<T, VEC, ARG>(arg, os, t0, a, parts);
      else load_arg<T, VEC, ARG>(arg, t0, a);
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        T cc[7];
#pragma unroll
        for (int k = 0; k < 7; k++) cc[k] = c[k][j];
        const T r = elem_1d_flags<T>(fn, MOREAU ? moreau_pre<T>(a[j], tau_scal, td[j], invert_tau) : a[j], elem_tau<T>(tau_scal, td[j], inner_inv), cc, e_zero, a_one);
        out[j] = MOREAU ? moreau_post<T>(a[j], r, tau_scal, td[j], invert_tau) : r;
      }
      stv<T, VEC>(res + t0, out);
      if constexpr (OPA) { if (with_res) residual_terms<T, VEC, ARG>(arg, parts, out, r_a, r_b); }
    } else {
      T v[D][VEC], tdv[MOREAU ? D : 1][VEC], norm[VEC], scale[VEC];
#pragma unroll
      for (int j = 0; j < VEC; j++) norm[j] = 0;
      ArgParts<T, OPA ? VEC : 1> parts[OPA ? D : 1];
      if (DIM > 0) {
#pragma unroll
        for (int i = 0; i < D; i++) {
          if constexpr (OPA) load_arg_op<T, VEC, ARG>(arg, os, t0 + (size_t)i * count, v[i], parts[i]);
          else load_arg<T, VEC, ARG>(arg, t0 + (size_t)i * count, v[i]);
          if (MOREAU) {
            if (i == 0) {
#pragma unroll
              for (int j = 0; j < VEC; j++) tdv[0][j] = td[j];
            } else ldv<T, VEC>(tau_diag + t0 + (size_t)i * count, tdv[MOREAU ? i : 0]);
          }
#pragma unroll
          for (int j = 0; j < VEC; j++) {
            const T sv = MOREAU ? moreau_pre<T>(v[i][j], tau_scal, tdv[MOREAU ? i : 0][j], invert_tau) : v[i][j];
            norm[j] += sv * sv;
          }
        }
      } else {
        for (size_t i = 0; i < dim; i++) {
          T w[VEC], tw[VEC];
          if constexpr (OPA) load_arg_op<T, VEC, ARG>(arg, os, t0 + i * count, w, parts[0], false);
          else load_arg<T, VEC, ARG>(arg, t0 + i * count, w);
          if (MOREAU) ldv<T, VEC>(tau_diag + t0 + i * count, tw);
#pragma unroll
          for (int j = 0; j < VEC; j++) {
            const T sv = MOREAU ? moreau_pre<T>(w[j], tau_scal, tw[j], invert_tau) : w[j];
            norm[j] += sv * sv;
          }
        }
      }
      bool pos[VEC];
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        pos[j] = norm[j] > 0;
        scale[j] = 0;
        if (pos[j]) {
          norm[j] = t_sqrt(norm[j]);
          T cc[7];
#pragma unroll
          for (int k = 0; k < 7; k++) cc[k] = c[k][j];
          scale[j] = scaled_prox_flags<T>(fn, norm[j], elem_tau<T>(tau_scal, td[j], inner_inv), cc, e_zero, a_one);
        }
      }
      if (DIM > 0) {
#pragma unroll
        for (int i = 0; i < D; i++) {
          T out[VEC];
#pragma unroll
          for (int j = 0; j < VEC; j++) {
            const T tdi = MOREAU ? tdv[MOREAU ? i : 0][j] : (T)0;
            const T sv = MOREAU ? moreau_pre<T>(v[i][j], tau_scal, tdi, invert_tau) : v[i][j];
            const T r = pos[j] ? scale[j] * sv / norm[j] : (T)0;
            out[j] = MOREAU ? moreau_post<T>(v[i][j], r, tau_scal, tdi, invert_tau) : r;
          }
          stv<T, VEC>(res + t0 + (size_t)i * count, out);
          if constexpr (OPA) { if (with_res) residual_terms<T, VEC, ARG>(arg, parts[i], out, r_a, r_b); }
        }
      } else {
        for (size_t i = 0; i < dim; i++) {
          T w[VEC], tw[VEC], out[VEC];
          if constexpr (OPA) load_arg_op<T, VEC, ARG>(arg, os, t0 + i * count, w, parts[0]);
          else load_arg<T, VEC, ARG>(arg, t0 + i * count, w);
          if (MOREAU) ldv<T, VEC>(tau_diag + t0 + i * count, tw);
#pragma unroll
          for (int j = 0; j < VEC; j++) {
            const T tdi = MOREAU ? tw[j] : (T)0;
            const T sv = MOREAU ? moreau_pre<T>(w[j], tau_scal, tdi, invert_tau) : w[j];
            const T r = pos[j] ? scale[j] * sv / norm[j] : (T)0;
            out[j] = MOREAU ? moreau_post<T>(w[j], r, tau_scal, tdi, invert_tau) : r;
          }
          stv<T, VEC>(res + t0 + i * count, out);
          if constexpr (OPA) { if (with_res) residual_terms<T, VEC, ARG>(arg, parts[0], out, r_a, r_b); }
        }
      }
    }
  }
  if constexpr (OPA) { if (with_res) block_dd_store2(r_a, r_b, os.res_ws, os.res_slot + blockIdx.x); }
}

template <class T, bool MOREAU, int ARG>
static int launch_prox_elem(int op, int fn, T* res, const ArgSrc<T>& arg, const T* tau_diag, double tau, int invert, size_t count,
                            size_t dim, int interleaved, const T* const* coeff_ptr, const double* coeff_val, void* stream,
                            const typename OpSrcOf<T, ARG>::type& os = typename OpSrcOf<T, ARG>::type(), unsigned max_grid = 0) {
  if (fn < 0 || fn >= PROST_FN_COUNT) { set_error("prox_elem: unknown function id"); return 1; }
  if (op != PROST_OP_1D && op != PROST_OP_NORM2) { set_error("prox_elem: unknown elem operation"); return 1; }
  if (count == 0) return 0;
  Coeffs<T> cf;
  for (int i = 0; i < 7; i++) { cf.ptr[i] = coeff_ptr ? coeff_ptr[i] : nullptr; cf.val[i] = (T)coeff_val[i]; }
  hipStream_t s = as_stream(stream);
  constexpr int V = VecOf<T>::N;
  bool vec = (op == PROST_OP_1D || !interleaved || dim == 1) && count % V == 0 && aligned16(res) && aligned16(arg.v0) && aligned16(tau_diag);
  if (ARG >= 1) vec = vec && aligned16(arg.v1);
  if (ARG == 1 || ARG == 2 || ARG == 5 || ARG == 6) vec = vec && aligned16(arg.v2);
  if (ARG == 2 || ARG == 3 || ARG == 5 || ARG == 6) vec = vec && aligned16(arg.v3);
  for (int i = 0; i < 7; i++) vec = vec && aligned16(cf.ptr[i]);
  if (vec) {
    const bool e_zero = !cf.ptr[4] && cf.val[4] == (T)0, a_one = !cf.ptr[0] && cf.val[0] == (T)1;
    unsigned gx = grid_for(count / V);
    if (max_grid && gx > max_grid) gx = max_grid;          // (residual sums: one partial slot per workgroup)
    dim3 g(gx), b(kBlock);
#define GO(OPv, DIMv) hipLaunchKernelGGL((prox_elem_vec_kernel<T, OPv, DIMv, MOREAU, ARG>), g, b, 0, s, res, arg, tau_diag, (T)tau, invert != 0, count, dim, fn, cf, e_zero, a_one, os)
    if (op == PROST_OP_1D) GO(PROST_OP_1D, 1);
    else if (dim == 1) GO(PROST_OP_NORM2, 1);
    else if (dim == 2) GO(PROST_OP_NORM2, 2);
    else if (dim == 3) GO(PROST_OP_NORM2, 3);
    else if (dim == 4) GO(PROST_OP_NORM2, 4);
    else GO(PROST_OP_NORM2, 0);
#undef GO
    PH_LAUNCH_END("prox_elem kernel");
  }
  if constexpr (ARG >= 3) { set_error("prox_elem_arg: the operator sources and the residual sums need the planar layout (or the 1-D operation), a count that is a multiple of 16 bytes and 16-byte aligned operands (prost_hip_prox_elem_arg_op_supported)"); return 1; }
  else {
  if (op == PROST_OP_1D)
    hipLaunchKernelGGL((prox_elem_kernel<T, PROST_OP_1D, MOREAU, ARG>), dim3(grid_for(count)), dim3(kBlock), 0, s, res, arg, tau_diag, (T)tau, invert != 0, count, (size_t)1, interleaved != 0, fn, cf);
  else
    hipLaunchKernelGGL((prox_elem_kernel<T, PROST_OP_NORM2, MOREAU, ARG>), dim3(grid_for(count)), dim3(kBlock), 0, s, res, arg, tau_diag, (T)tau, invert != 0, count, dim, interleaved != 0, fn, cf);
  PH_LAUNCH_END("prox_elem kernel");
  }
}

template <class T, bool MOREAU>
static int launch_prox_elem(int op, int fn, T* res, const T* arg, const T* tau_diag, double tau, int invert, size_t count,
                            size_t dim, int interleaved, const T* const* coeff_ptr, const double* coeff_val, void* stream) {
  return launch_prox_elem<T, MOREAU, 0>(op, fn, res, ArgSrc<T>{arg, nullptr, nullptr, nullptr, (T)0, (T)0, nullptr}, tau_diag, tau, invert, count, dim, interleaved,
                                        coeff_ptr, coeff_val, stream);
}

template <class T>
static int launch_prox_elem_arg(int op, int fn, int moreau, T* res, const prost_hip_arg_spec* a, const T* tau_diag, double tau, int invert, size_t count,
                                size_t dim, int interleaved, const T* const* coeff_ptr, const double* coeff_val, void* stream) {
  if (!a) { set_error("prox_elem_arg: argument source required"); return 1; }
  const ArgSrc<T> src{static_cast<const T*>(a->v[0]), static_cast<const T*>(a->v[1]), static_cast<const T*>(a->v[2]), static_cast<const T*>(a->v[3]),
                      (T)a->s[0], (T)a->s[1], a->mode == PROST_ARG_PLAIN ? nullptr : step_record<T>()};
  if (src.v0 == res) { set_error("prox_elem_arg: the result must not alias the argument source"); return 1; }
  if ((a->mode == PROST_ARG_PDHG_PRIMAL || a->mode == PROST_ARG_PDHG_DUAL) && a->res_ws) {
    // the residual terms of this prox's elements are added up by the launch itself (ARG 5 / 6): 16-bytes-per-lane kernel only
    if (a->res_slots_max < 1) { set_error("prox_elem_arg: no partial slots for the residual sums"); return 1; }
    OpSrc<T> os;
    os.opp = nullptr; os.base = 0; os.w0 = os.w1 = nullptr; os.kty_out = nullptr; os.use0 = os.use1 = 0;
    os.res_ws = a->res_ws; os.res_slot = a->res_slot;
    const unsigned cap = a->res_slots_max;
    if (a->mode == PROST_ARG_PDHG_PRIMAL) {
      if (moreau) return launch_prox_elem<T, true, 5>(op, fn, res, src, tau_diag, tau, invert, count, dim, interleaved, coeff_ptr, coeff_val, stream, os, cap);
      return launch_prox_elem<T, false, 5>(op, fn, res, src, tau_diag, tau, invert, count, dim, interleaved, coeff_ptr, coeff_val, stream, os, cap);
    }
    if (moreau) return launch_prox_elem<T, true, 6>(op, fn, res, src, tau_diag, tau, invert, count, dim, interleaved, coeff_ptr, coeff_val, stream, os, cap);
    return launch_prox_elem<T, false, 6>(op, fn, res, src, tau_diag, tau, invert, count, dim, interleaved, coeff_ptr, coeff_val, stream, os, cap);
  }
#define GO(M, A) return launch_prox_elem<T, M, A>(op, fn, res, src, tau_diag, tau, invert, count, dim, interleaved, coeff_ptr, coeff_val, stream)
  switch (a->mode) {
    case PROST_ARG_PLAIN: if (moreau) GO(true, 0); else GO(false, 0);
    case PROST_ARG_PDHG_PRIMAL: if (moreau) GO(true, 1); else GO(false, 1);
    case PROST_ARG_PDHG_DUAL: if (moreau) GO(true, 2); else GO(false, 2);
    case PROST_ARG_PDHG_PRIMAL_OP: case PROST_ARG_PDHG_DUAL_OP: break;
    default: set_error("prox_elem_arg: unknown argument mode"); return 1;
  }
#undef GO
  // operator sources: K^T y / K x, K x_prev formed on the fly from the blocks of a->op (fused_op.hpp)
  constexpr unsigned V = VecOf<T>::N;
  const bool primal = a->mode == PROST_ARG_PDHG_PRIMAL_OP;
  if (!a->op || !a->w[0] || (!primal && !a->w[1])) { set_error("prox_elem_arg: the operator sources need the operator and the whole vectors"); return 1; }
  if (!fused_op_ok(a->op, a->op_rows, a->op_cols) || !fused_op_vec_ok(a->op, V) || a->base % V) { set_error("prox_elem_arg: unsupported operator description (prost_hip_prox_elem_arg_op_supported)"); return 1; }
  if (a->base + count * (op == PROST_OP_1D ? 1 : dim) > (primal ? a->op_cols : a->op_rows)) { set_error("prox_elem_arg: the prox's range exceeds the operator"); return 1; }
  OpSrc<T> os;
  os.opp = device_op(a->op); os.base = (size_t)a->base;
  if (!os.opp) return 1;
  os.w0 = static_cast<const T*>(a->w[0]); os.w1 = static_cast<const T*>(a->w[1]);
  os.kty_out = static_cast<T*>(a->kty_out); os.use0 = a->use[0]; os.use1 = a->use[1];
  os.res_ws = a->res_ws; os.res_slot = a->res_slot;
  if (os.res_ws && a->res_slots_max < 1) { set_error("prox_elem_arg: no partial slots for the residual sums"); return 1; }
  if (os.kty_out && !aligned16(os.kty_out)) { set_error("prox_elem_arg: kty_out must be 16-byte aligned"); return 1; }
  const unsigned cap = os.res_ws ? a->res_slots_max : 0;
  if (primal) {
    if (moreau) return launch_prox_elem<T, true, 3>(op, fn, res, src, tau_diag, tau, invert, count, dim, interleaved, coeff_ptr, coeff_val, stream, os, cap);
    return launch_prox_elem<T, false, 3>(op, fn, res, src, tau_diag, tau, invert, count, dim, interleaved, coeff_ptr, coeff_val, stream, os, cap);
  }
  if (moreau) return launch_prox_elem<T, true, 4>(op, fn, res, src, tau_diag, tau, invert, count, dim, interleaved, coeff_ptr, coeff_val, stream, os, cap);
  return launch_prox_elem<T, false, 4>(op, fn, res, src, tau_diag, tau, invert, count, dim, interleaved, coeff_ptr, coeff_val, stream, os, cap);
}

// ------------------------------------------------------------------------------------------
// ProxIndEpiQuad (prox_ind_epi_quad.cu:42-79) with helper::ProjectEpiQuadNd (helper.hpp:44-105)
// planar layout: x_i at tx + count*i (i < dim-1), y at tx + count*(dim-1)
// ------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kBlock) epi_quad_kernel(T* __restrict__ res, const T* __restrict__ arg, size_t count,
                                                          size_t dim, const T* __restrict__ a_ptr, T a_val,
                                                          const T* __restrict__ b_ptr, const T* __restrict__ c_ptr, T c_val) {
  for (size_t tx = (size_t)blockIdx.x * kBlock + threadIdx.x; tx < count; tx += (size_t)gridDim.x * kBlock) {
    const size_t d = dim - 1;
    const T a = a_ptr ? a_ptr[tx] : a_val;
    const T c = c_ptr ? c_ptr[tx] : c_val;
    T sq_norm_b = 0;
    for (size_t i = 0; i < d; i++) {
      const T val = b_ptr[tx + count * i];
      res[tx + count * i] = arg[tx + count * i] + (val / (2 * a));
      sq_norm_b += val * val;
    }
    const T y0 = arg[count * d + tx] - c + (sq_norm_b / (4 * a));
    // ---- ProjectEpiQuadNd(x, y0, alpha = a, x, y, d) ----
    T sq_norm_x0 = 0;
    for (size_t i = 0; i < d; i++) { const T v = res[tx + count * i]; sq_norm_x0 += v * v; }
    const T norm_x0 = t_sqrt(sq_norm_x0);
    T y;
    if (y0 >= a * sq_norm_x0) {
      y = y0;
    } else {
      const T pa = (T)(2. * (double)a * (double)norm_x0);
      const T pb = (T)(2. * (1. - 2. * (double)a * (double)y0) / 3.);
      T dd, v;
      if (pb < 0) {
        const T sq = t_pow(-pb, (T)(3. / 2.));
        dd = (pa - sq) * (pa + sq);
      } else {
        dd = pa * pa + pb * pb * pb;
      }
      if (dd >= 0) {
        const T cc = t_pow(pa + t_sqrt(dd), (T)(1. / 3.));
        if ((double)t_abs(cc) > 1e-6) v = cc - pb / cc; else v = 0;
      } else {
        v = 2 * t_sqrt(-pb) * t_cos(t_acos(pa / t_pow(-pb, (T)(3. / 2.))) / (T)3.);
      }
      if (norm_x0 > 0) {
        for (size_t i = 0; i < d; i++)
          res[tx + count * i] = (T)(((double)v / (2. * (double)a)) * (double)(res[tx + count * i] / norm_x0));
      } else {
        for (size_t i = 0; i < d; i++) res[tx + count * i] = 0;
      }
      T sq_norm_x = 0;
      for (size_t i = 0; i < d; i++) { const T w = res[tx + count * i]; sq_norm_x += w * w; }
      y = a * sq_norm_x;
    }
    for (size_t i = 0; i < d; i++) res[tx + count * i] -= b_ptr[tx + count * i] / (2 * a);
    res[count * d + tx] = y + c - (sq_norm_b / (4 * a));
  }
}

// ------------------------------------------------------------------------------------------
// Moreau pre/post scaling (prox_moreau.cu:29-61)
// ------------------------------------------------------------------------------------------
// in = arg, tau_diag
template <class T> struct MoreauPreF {
  T tau; bool inv;
  __device__ T operator()(const T* a) const { return inv ? a[0] * (tau * a[1]) : a[0] / (tau * a[1]); }
};
// in = arg, tau_diag, res (in place)
template <class T> struct MoreauPostF {
  T tau; bool inv;
  __device__ T operator()(const T* a) const { return inv ? a[0] - a[2] / (tau * a[1]) : a[0] - tau * a[1] * a[2]; }
};

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_prox_elem_f32(int op, int fn, float* res, const float* arg, const float* td, double tau, int inv, size_t count, size_t dim, int il, const float* const* cp, const double* cv, void* s) {
  return launch_prox_elem<float, false>(op, fn, res, arg, td, tau, inv, count, dim, il, cp, cv, s);
}
int prost_hip_prox_elem_f64(int op, int fn, double* res, const double* arg, const double* td, double tau, int inv, size_t count, size_t dim, int il, const double* const* cp, const double* cv, void* s) {
  return launch_prox_elem<double, false>(op, fn, res, arg, td, tau, inv, count, dim, il, cp, cv, s);
}
int prost_hip_prox_elem_moreau_f32(int op, int fn, float* res, const float* arg, const float* td, double tau, int inv, size_t count, size_t dim, int il, const float* const* cp, const double* cv, void* s) {
  return launch_prox_elem<float, true>(op, fn, res, arg, td, tau, inv, count, dim, il, cp, cv, s);
}
int prost_hip_prox_elem_moreau_f64(int op, int fn, double* res, const double* arg, const double* td, double tau, int inv, size_t count, size_t dim, int il, const double* const* cp, const double* cv, void* s) {
  return launch_prox_elem<double, true>(op, fn, res, arg, td, tau, inv, count, dim, il, cp, cv, s);
}
int prost_hip_prox_elem_arg_op_supported(const prost_hip_fused_op* op, uint64_t m, uint64_t n, int dtype) {
  return fused_op_ok(op, m, n) && fused_op_vec_ok(op, dtype == 0 ? 4u : 2u) ? 1 : 0;
}
int prost_hip_prox_elem_arg_f32(int op, int fn, int moreau, float* res, const prost_hip_arg_spec* a, const float* td, double tau, int inv, size_t count, size_t dim, int il, const float* const* cp, const double* cv, void* s) {
  return launch_prox_elem_arg<float>(op, fn, moreau, res, a, td, tau, inv, count, dim, il, cp, cv, s);
}
int prost_hip_prox_elem_arg_f64(int op, int fn, int moreau, double* res, const prost_hip_arg_spec* a, const double* td, double tau, int inv, size_t count, size_t dim, int il, const double* const* cp, const double* cv, void* s) {
  return launch_prox_elem_arg<double>(op, fn, moreau, res, a, td, tau, inv, count, dim, il, cp, cv, s);
}
int prost_hip_prox_epi_quad_f32(float* res, const float* arg, size_t count, size_t dim, const float* a_ptr, double a_val, const float* b_ptr, const float* c_ptr, double c_val, void* s) {
  if (count == 0) return 0;
  hipLaunchKernelGGL((epi_quad_kernel<float>), dim3(grid_for(count)), dim3(kBlock), 0, as_stream(s), res, arg, count, dim, a_ptr, (float)a_val, b_ptr, c_ptr, (float)c_val);
  PH_LAUNCH_END("epi_quad kernel");
}
int prost_hip_prox_epi_quad_f64(double* res, const double* arg, size_t count, size_t dim, const double* a_ptr, double a_val, const double* b_ptr, const double* c_ptr, double c_val, void* s) {
  if (count == 0) return 0;
  hipLaunchKernelGGL((epi_quad_kernel<double>), dim3(grid_for(count)), dim3(kBlock), 0, as_stream(s), res, arg, count, dim, a_ptr, a_val, b_ptr, c_ptr, c_val);
  PH_LAUNCH_END("epi_quad kernel");
}
int prost_hip_moreau_prescale_f32(float* o, const float* a, const float* td, double tau, int inv, size_t n, void* s) {
  return launch_ew<float, 2>("moreau prescale", o, EwIn<float, 2>{{a, td}}, n, MoreauPreF<float>{(float)tau, inv != 0}, as_stream(s));
}
int prost_hip_moreau_prescale_f64(double* o, const double* a, const double* td, double tau, int inv, size_t n, void* s) {
  return launch_ew<double, 2>("moreau prescale", o, EwIn<double, 2>{{a, td}}, n, MoreauPreF<double>{tau, inv != 0}, as_stream(s));
}
int prost_hip_moreau_postscale_f32(float* r, const float* a, const float* td, double tau, int inv, size_t n, void* s) {
  return launch_ew<float, 3>("moreau postscale", r, EwIn<float, 3>{{a, td, r}}, n, MoreauPostF<float>{(float)tau, inv != 0}, as_stream(s));
}
int prost_hip_moreau_postscale_f64(double* r, const double* a, const double* td, double tau, int inv, size_t n, void* s) {
  return launch_ew<double, 3>("moreau postscale", r, EwIn<double, 3>{{a, td, r}}, n, MoreauPostF<double>{tau, inv != 0}, as_stream(s));
}
}  // extern "C"
