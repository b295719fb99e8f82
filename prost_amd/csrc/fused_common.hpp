// fused_common.hpp -- helpers shared by the fused PDHG kernels (kernels_fused*.hip).
#pragma once
#include "common.hpp"
#include "device_math.hpp"

namespace prost_hip {

// a lane owns 16 bytes of consecutive rows: float4 / double2
template <class T> struct VecOf;
typedef float native_f4 __attribute__((ext_vector_type(4)));
typedef double native_d2 __attribute__((ext_vector_type(2)));
template <> struct VecOf<float> { static constexpr int N = 4; typedef float4 type; typedef native_f4 native; };
template <> struct VecOf<double> { static constexpr int N = 2; typedef double2 type; typedef native_d2 native; };

template <class T, int VEC>
__device__ __forceinline__ void ldv(const T* __restrict__ p, T (&v)[VEC]) {
  if (VEC == 1) { v[0] = p[0]; return; }
  typedef typename VecOf<T>::type V;
  const V t = *reinterpret_cast<const V*>(p);
  const T* e = reinterpret_cast<const T*>(&t);
#pragma unroll
  for (int j = 0; j < VEC; j++) v[j] = e[j];
}
template <class T, int VEC>
__device__ __forceinline__ void stv(T* __restrict__ p, const T (&v)[VEC]) {
  if (VEC == 1) { p[0] = v[0]; return; }
  typedef typename VecOf<T>::type V;
  V t;
  T* e = reinterpret_cast<T*>(&t);
#pragma unroll
  for (int j = 0; j < VEC; j++) e[j] = v[j];
  *reinterpret_cast<V*>(p) = t;
}

// streaming (non-temporal) forms: the PDHG vectors are touched once per iteration and the working
// set (470 MB at 4096^2) exceeds every cache level
template <class T, int VEC>
__device__ __forceinline__ void ldv_nt(const T* __restrict__ p, T (&v)[VEC]) {
  if (VEC == 1) { v[0] = __builtin_nontemporal_load(p); return; }
  typedef typename VecOf<T>::native V;
  const V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#pragma unroll
  for (int j = 0; j < VEC; j++) v[j] = t[j];
}
template <class T, int VEC>
__device__ __forceinline__ void stv_nt(T* __restrict__ p, const T (&v)[VEC]) {
  if (VEC == 1) { __builtin_nontemporal_store(v[0], p); return; }
  typedef typename VecOf<T>::native V;
  V t;
#pragma unroll
  for (int j = 0; j < VEC; j++) t[j] = v[j];
  __builtin_nontemporal_store(t, reinterpret_cast<V*>(p));
}

// ---- image heights that are not a multiple of VEC ------------------------------------------------
// Columns then start at addresses that are only sizeof(T)-aligned and the last lane of a column holds
// fewer than VEC valid rows.  gfx950 serves 16-byte global accesses at any 4-byte aligned address
// (unaligned access mode), so the full vectors stay ONE load / store each; only the lane with the
// ragged end falls back to element accesses (it must neither read past the allocation nor write
// into the next column).  nvalid = number of rows of this lane inside the image (VEC for all but one).
typedef float native_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef double native_d2u __attribute__((ext_vector_type(2), aligned(8)));
template <class T> struct UVecOf;
template <> struct UVecOf<float> { typedef native_f4u type; };
template <> struct UVecOf<double> { typedef native_d2u type; };

template <class T, int VEC, bool RAG>
__device__ __forceinline__ void ldv_n(const T* p, T (&v)[VEC], int nvalid) {
  if (VEC == 1) { v[0] = p[0]; return; }
  if (!RAG) { ldv<T, VEC>(p, v); return; }                   // heights that are a multiple of VEC: the aligned form, unchanged
  if (nvalid >= VEC) {
    typedef typename UVecOf<T>::type V;
    const V t = *reinterpret_cast<const V*>(p);
#pragma unroll
    for (int j = 0; j < VEC; j++) v[j] = t[j];
  } else {
#pragma unroll
    for (int j = 0; j < VEC; j++) v[j] = j < nvalid ? p[j] : (T)0;
  }
}
template <class T, int VEC, bool NT, bool RAG>
__device__ __forceinline__ void stv_n(T* p, const T (&v)[VEC], int nvalid) {
  if (VEC == 1) { if (NT) __builtin_nontemporal_store(v[0], p); else p[0] = v[0]; return; }
  if (!RAG) { if (NT) stv_nt<T, VEC>(p, v); else stv<T, VEC>(p, v); return; }
  if (nvalid >= VEC) {
    typedef typename UVecOf<T>::type V;
    V t;
#pragma unroll
    for (int j = 0; j < VEC; j++) t[j] = v[j];
    if (NT) __builtin_nontemporal_store(t, reinterpret_cast<V*>(p)); else *reinterpret_cast<V*>(p) = t;
  } else {
#pragma unroll
    for (int j = 0; j < VEC; j++) if (j < nvalid) p[j] = v[j];
  }
}

// uniform base + 32-bit per-lane byte offset (global_load/store saddr form)
template <class T, int VEC, bool RAG>
__device__ __forceinline__ void ldv_o(const T* base, unsigned byte_off, T (&v)[VEC], int nvalid) {
  ldv_n<T, VEC, RAG>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off), v, nvalid);
}
template <class T, int VEC, bool NT, bool RAG>
__device__ __forceinline__ void stv_o(T* base, unsigned byte_off, const T (&v)[VEC], int nvalid) {
  stv_n<T, VEC, NT, RAG>(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off), v, nvalid);
}

template <class T>
struct FusedArgs {
  size_t nx, ny, L;
  int cols_per_block;
  unsigned chunks;            // column chunks per row strip (single-kernel iteration: 1-D tile grid)
  int g_fn, f_fn;
  const T* g_ptr[7]; T g_val[7];
  const T* f_ptr[7]; T f_val[7];
  T Tval, Sval;
  size_t rx0, rx1;            // columns whose residual terms are counted (single-kernel / pair kernels)
  int varT;                   // position-dependent primal preconditioner (prost_hip_fused_desc.var_T): Tcls by stencil entries per column
  T Tcls[3];
  int fmor;                   // prox_fstar = Moreau wrap of the described norm2 operation (prost_hip_fused_desc.f_moreau)
  unsigned strips;            // K-iteration kernel: row strips of the launch (tile order: vertical neighbours are consecutive tiles)
};

// step size the dual elem operation sees: sigma Sigma, or -- inside a Moreau wrap, which calls it with the inverted flag --
// 1 / (sigma Sigma) as ElemOperation forms it (elem_operation_1d.hpp:40: 1. / (tau_scal * tau_diag) in double, narrowed)
template <class T>
__host__ __device__ inline T dual_prox_step(T sigma, T Sval, int fmor) { return fmor ? (T)(1. / (double)(sigma * Sval)) : sigma * Sval; }

// step sizes of one iteration + the element-independent prox terms that go with them (device_math.hpp: UniformProx)
// the step-size dependent terms of prox_g for the pixels whose Tau_j differs from the interior's (FusedArgs::varT): the step
// c a^2 tau Tau_j and the divisor 1 + step of Function1DSquare with its reciprocal, per class (2 / 3 stencil entries per column)
template <class T>
struct EdgeTerms {
  UniformDiv sq;
  T step;
};
template <class T>
__host__ __device__ inline EdgeTerms<T> make_edge_terms(const T* g_val, T tauT) {
  const UniformProx<T> u = make_uniform_prox<T>(g_val, tauT);
  EdgeTerms<T> e;
  e.sq = u.sq; e.step = u.step;
  return e;
}
template <class T>
struct IterParams {
  T tau, sigma, theta;
  UniformProx<T> ug, uf;
  EdgeTerms<T> ec[2];          // varT only: classes Tcls[0] (corner), Tcls[1] (edge)
};

// Device-resident step sizes for the residual-driven rules (goldstein, boyd; kernels_pdhg_rule.hip).  The reference forms
// the four residual norms on the host and adapts tau / sigma there (backend_pdhg.cu:433-476) -- a device-to-host round trip
// per residual iteration, i.e. per iteration under its default options (pdhg.m:4-14: boyd, residual_iter = 1).  Here the rule
// is evaluated by a one-thread kernel behind the reduction of the sums; it leaves the parameters of the NEXT iterations in `p`,
// which the iteration kernels read through scalar loads (a wave-uniform record, fetched once per wavefront), and raises `stop`
// when the solver's stopping test fires: every later iteration kernel and rule evaluation of the same batch then returns at
// once, so the host can enqueue a batch of iterations without looking at the device in between.
template <class T>
struct PdhgRecord {
  IterParams<T> p;
  int stop;                          // the stopping test of solver.cu:141-150 fired (only raised when stop_on_convergence)
  int stop_on_convergence;
  int variant;                       // PROST_PDHG_RULE_*
  int arb_l, arb_u;
  T arg_alpha, arg_nu, arg_delta, arb_delta, arb_tau;
  T tol_abs_primal, tol_abs_dual, tol_rel_primal, tol_rel_dual;
  double sqrt_rows, sqrt_cols;       // sqrt of the GLOBAL sizes (backend.hpp:71-74)
  T g_val[7], f_val[7], Tval, Sval;  // what make_uniform_prox needs beside the step size
  int varT;                          // FusedArgs::varT and the two other classes of Tau_j
  T Tcls[2];
  int fmor;                          // FusedArgs::fmor
  unsigned long long evaluations;    // rule evaluations since prost_hip_pdhg_rule_begin
  unsigned long long stop_iteration;
};

// the record the generic PDHG kernels launched by this thread take their step sizes from, or null (prost_hip_use_step_record)
template <class T> inline const PdhgRecord<T>* step_record() { return static_cast<const PdhgRecord<T>*>(g_step_record); }

// value of the row above the first row of this lane (row0 - 1): neighbour lane's last element
template <class T, int VEC>
__device__ __forceinline__ T row_above(const T (&v)[VEC], const T* __restrict__ col_base, size_t row0, bool active) {
  T up = lane_up(v[VEC - 1]);
  if ((threadIdx.x & (kWave - 1)) == 0 && active && row0 > 0) up = col_base[row0 - 1];
  return up;
}
// value of the row below the last row of this lane (row0 + VEC)
template <class T, int VEC>
__device__ __forceinline__ T row_below(const T (&v)[VEC], const T* __restrict__ col_base, size_t row0, size_t ny, bool active) {
  T dn = lane_down(v[0]);
  if ((threadIdx.x & (kWave - 1)) == kWave - 1 && active && row0 + VEC < ny) dn = col_base[row0 + VEC];
  return dn;
}

// the mask sentinel of a merged b stream (prost_hip_mask_merge): bit compare, one instruction
__device__ __forceinline__ bool is_mask_sentinel(float b) { return __float_as_uint(b) == PROST_HIP_MASK_SENTINEL_F32; }
__device__ __forceinline__ bool is_mask_sentinel(double b) { return (unsigned long long)__double_as_longlong(b) == PROST_HIP_MASK_SENTINEL_F64; }

template <class T>
inline FusedArgs<T> make_fused_args(const prost_hip_fused_desc* d) {
  FusedArgs<T> a;
  a.nx = d->nx; a.ny = d->ny; a.L = d->L; a.g_fn = d->g_fn; a.f_fn = d->f_fn;
  for (int k = 0; k < 7; k++) {
    a.g_ptr[k] = static_cast<const T*>(d->g_coeff_ptr[k]); a.g_val[k] = (T)d->g_coeff_val[k];
    a.f_ptr[k] = static_cast<const T*>(d->f_coeff_ptr[k]); a.f_val[k] = (T)d->f_coeff_val[k];
  }
  a.Tval = (T)d->T_val; a.Sval = (T)d->S_val;
  a.rx0 = d->res_x1 ? d->res_x0 : 0; a.rx1 = d->res_x1 ? d->res_x1 : d->nx;
  a.cols_per_block = 16;
  a.varT = d->var_T ? 1 : 0;
  for (int k = 0; k < 3; k++) a.Tcls[k] = (T)d->T_cls[k];
  a.fmor = d->f_moreau ? 1 : 0;
  a.strips = 0;
  return a;
}

// gradient3d passes (kernels_fused3d.hip)
bool fused3d_desc_ok(const prost_hip_fused_desc* d);
template <class T> int run_primal3d(const prost_hip_fused_desc* d, T* x_new, const T* x, const T* y, const T* y_prev, double tau, int use_kty, int use_kty_prev, double* out2, void* ws, void* stream);
template <class T> int run_dual3d(const prost_hip_fused_desc* d, T* y_new, const T* y, const T* xn, const T* xo, double sigma, double theta, int use_kx_prev, double* out2, void* ws, void* stream);

// folds nslots x 4 doubles (one partial per wavefront) in a fixed order -> out4 (kernels_fused_iter.hip)
int launch_fold4(double* out4, const double* partial, unsigned nslots, hipStream_t s);
// ... with the step-size rule and the stopping test of the device record `rec` (PdhgRecord<T>) applied to the sums in the same launch
struct RuleTail { int apply; unsigned long long iteration; prost_hip_pdhg_rule_state* mirror; };
template <class T>
int launch_fold4_rule(double* out4, const double* partial, unsigned nslots, void* rec, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, hipStream_t s);

inline bool aligned16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace prost_hip
