// kernels_cgls.hip -- device-resident CGLS for the ADMM graph projection on gfx950.
//
// Follows cgls::Solve (reference include/prost/cgls.hpp:222-371) applied to the preconditioned
// operator A = Sigma^(1/2) K Tau^(1/2) of GemvPrecondK (src/backend/backend_admm.cu:199-272).
// The reference drives the solve from the host: every nrm2 is a device reduction followed by a
// blocking copy of one scalar (4 per CG iteration), and every vector update is its own pass.
// Here the CG scalars (gamma, alpha, beta, the norms, the stopping flag) live in a small device
// record; each stage below is ONE pass that fuses the elementwise work between two operator
// applications with the reductions it feeds (one partial sum per workgroup, reduce.hpp), and a
// one-workgroup scalar kernel folds the partials in a fixed order and advances the scalar
// recurrence -- two of those per CG iteration (alpha; beta + stopping test), no host round trip
// inside the solve, run-to-run deterministic.  Once the stopping test fires the remaining stages
// return immediately (the host may keep launching them).
//
// Tried and dropped: folding in the last workgroup to retire (ticket + __threadfence).  On gfx950 the
// agent-scope release is an L2 write-back per workgroup in the middle of a streaming kernel; the
// stages ran 4-8x slower (StepXR 111 us vs 15 us at m = 5 Mi, n = 2 Mi).  Kernel boundaries are
// the cheap coherence point.
//
// Per-element arithmetic is the reference's expression order (-ffp-contract=off):
//   gemv_functor1  sqrt(d) v              gemv_functor2  (beta / (alpha sqrt(d))) v
//   gemv_functor3  alpha sqrt(d) v        axpy           alpha x + y
#include <limits>

#include <cstdlib>
#include "elementwise.hpp"
#include "fused_op.hpp"

namespace prost_hip {

struct CgState {
  double gamma, norms0, norms, normx, xmax, alpha, neg_alpha, beta;
  double tol;                      // stopping tolerance and epoch of the current solve: set by INIT_X, so that the
  int epoch;                       // STEP launches take no per-solve argument (they can be replayed from a HIP graph)
  int done, k, indefinite, flag;
};

enum { kRegionX = 0, kRegionS, kRegionP, kRegionQ, kRegions };          // partial-sum regions of the workspace
// a region holds one slot per workgroup: (hi, lo) of ONE sum, compact (stride 2: kernels with a single sum), or
// (a.hi, a.lo, b.hi, b.lo) (stride 4: the staged kernels and the residual stages, which carry two sums) -- reduce.hpp, dd_t
__host__ __device__ inline double* region(double* ws, int r) { return ws + (size_t)r * 4 * kReduceBlocks; }

template <class T, int VEC, class F>
__global__ void __launch_bounds__(kBlock) cg_stage_kernel(F f, size_t n0, size_t n1, const CgState* st, double* ws) {
  if (F::kSkipWhenDone && st->done) return;
  f.load(st);
  dd_t sa{0.0, 0.0}, sb{0.0, 0.0};
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x, stride = (size_t)gridDim.x * kBlock;
  {
    const size_t nv = n0 / VEC;
    for (size_t i = tid; i < nv; i += stride) f.template range0<VEC>(i * VEC, sa, sb);
    if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < n0 - nv * VEC) f.template range0<1>(nv * VEC + threadIdx.x, sa, sb);
  }
  if (F::kTwoRanges) {
    const size_t nv = n1 / VEC;
    for (size_t i = tid; i < nv; i += stride) f.template range1<VEC>(i * VEC, sa, sb);
    if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < n1 - nv * VEC) f.template range1<1>(nv * VEC + threadIdx.x, sa, sb);
  }
  if (F::kRegion >= 0) {
    block_dd_store2(sa, sb, region(ws, F::kRegion), blockIdx.x);
    if (F::kRegion2 >= 0 && threadIdx.x == 0) {
      region(ws, F::kRegion2)[4 * blockIdx.x] = region(ws, F::kRegion)[4 * blockIdx.x];
      region(ws, F::kRegion2)[4 * blockIdx.x + 1] = region(ws, F::kRegion)[4 * blockIdx.x + 1];
    }
  }
}

enum { kScalarsInitX = 0, kScalarsInitS, kScalarsAlpha, kScalarsBeta };
struct ScalarArgs { double shift, tol, eps; unsigned g0, g1; int* host_done; int epoch; unsigned stride; };   // stride (doubles per slot) 4: two-sum slots, 2: compact

// the scalar recurrences of cgls.hpp, in double, values narrowed to T where the reference narrows them
template <class T, int WHICH>
__global__ void __launch_bounds__(kBlock) cg_scalar_kernel(CgState* stp, const double* ws, ScalarArgs a) {
  if (WHICH >= kScalarsAlpha && stp->done) return;
  const double s0 = fold_dd(region(const_cast<double*>(ws), WHICH == kScalarsInitX ? kRegionX : WHICH == kScalarsAlpha ? kRegionQ : kRegionS), a.g0, a.stride);
  const double s1 = WHICH == kScalarsAlpha ? fold_dd(region(const_cast<double*>(ws), kRegionP), a.g1, 4)
                  : WHICH == kScalarsBeta ? fold_dd(region(const_cast<double*>(ws), kRegionX), a.g1, 4) : 0.;
  if (threadIdx.x != 0) return;
  CgState& st = *stp;
  if (WHICH == kScalarsInitX) {                        // cgls.hpp:243-249, :281-282
    st.normx = sqrt(s0); st.xmax = st.normx;
    st.done = 0; st.k = 0; st.indefinite = 0; st.flag = 0;
    st.tol = a.tol; st.epoch = a.epoch;
  } else if (WHICH == kScalarsInitS) {                 // :263-283
    st.norms = sqrt(s0); st.norms0 = st.norms;
    st.gamma = st.norms0 * st.norms0;
    if (st.norms < a.eps) { st.flag = 1; st.done = 1; }
  } else if (WHICH == kScalarsAlpha) {                 // :297-310
    const double normq = sqrt(s0), normp = sqrt(s1);
    double dlt = normq * normq + a.shift * normp * normp;
    if (dlt <= 0.) st.indefinite = 1;
    if (dlt == 0.) dlt = a.eps;
    st.alpha = (double)(T)(st.gamma / dlt);
    st.neg_alpha = (double)(T)(-st.gamma / dlt);
  } else {                                             // :326-360
    st.norms = sqrt(s0);
    const double gamma1 = st.gamma;
    st.gamma = st.norms * st.norms;
    st.beta = (double)(T)(st.gamma / gamma1);
    st.normx = sqrt(s1);
    st.xmax = st.xmax > st.normx ? st.xmax : st.normx;
    if ((st.norms <= st.norms0 * st.tol) || (st.normx * st.tol >= 1.)) {
      st.done = 1;
      if (a.host_done) __hip_atomic_store(a.host_done, st.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      st.k = st.k + 1;
    }
  }
}

// ---- stages -------------------------------------------------------------------------------------
// INIT_X (n):  t = sqrt(Tau) x [gemv1 of r = b - A x];  s = (-shift / (1 sqrt(Tau))) x [gemv2 of s = A'r - shift x,
//              applied to the copy s = x];  partials of |x|^2  (cgls.hpp:243-262)
template <class T> struct InitX {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = kRegionX, kRegion2 = -1;
  const T* x; const T* tau; T* t; T* s; T negshift;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t& sa, dd_t&) const {
    T xv[V], dv[V], tv[V], sv[V];
    ldv<T, V>(x + i, xv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(dv[j]);
      tv[j] = sq * xv[j];
      sv[j] = (negshift / ((T)1 * sq)) * xv[j];
      dd_acc(sa, (double)xv[j] * (double)xv[j]);
    }
    stv<T, V>(t + i, tv); stv<T, V>(s + i, sv);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// INIT_R (m):  r = (1 / (-1 sqrt(Sigma))) b   [gemv2 of r = b - A x on the copy r = b]
template <class T> struct InitR {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* b; const T* sigma; T* r;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t&, dd_t&) const {
    T bv[V], dv[V], rv[V];
    ldv<T, V>(b + i, bv); ldv<T, V>(sigma + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) rv[j] = ((T)1 / ((T)-1 * t_sqrt(dv[j]))) * bv[j];
    stv<T, V>(r + i, rv);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// INIT_R2 (m), after r += K t:  r = normx > 0 ? -1 sqrt(Sigma) r : b  (the reference skips the product for
//              x = 0, cgls.hpp:250);  t = sqrt(Sigma) r  [gemv1 of s = A'r - shift x]
template <class T> struct InitR2 {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* b; const T* sigma; T* r; T* t; bool nonzero;
  __device__ void load(const CgState* st) { nonzero = st->normx > 0.; }
  template <int V> __device__ void range0(size_t i, dd_t&, dd_t&) const {
    T bv[V], dv[V], rv[V], tv[V];
    ldv<T, V>(b + i, bv); ldv<T, V>(sigma + i, dv); ldv<T, V>(r + i, rv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(dv[j]);
      rv[j] = nonzero ? (T)-1 * sq * rv[j] : bv[j];
      tv[j] = sq * rv[j];
    }
    stv<T, V>(r + i, rv); stv<T, V>(t + i, tv);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// INIT_S (n), after s += K^T t:  s = 1 sqrt(Tau) s;  p = s;  t = sqrt(Tau) p [gemv1 of q = A p];
//              partials of |s|^2 = |p|^2  (cgls.hpp:263-283)
template <class T> struct InitS {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = kRegionS, kRegion2 = kRegionP;
  T* s; T* p; T* t; const T* tau;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t& sa, dd_t&) const {
    T sv[V], dv[V], tv[V];
    ldv<T, V>(s + i, sv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(dv[j]);
      sv[j] = (T)1 * sq * sv[j];
      tv[j] = sq * sv[j];
      dd_acc(sa, (double)sv[j] * (double)sv[j]);
    }
    stv<T, V>(s + i, sv); stv<T, V>(p + i, sv); stv<T, V>(t + i, tv);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// STEP_Q (m), after q = K t:  q = 1 sqrt(Sigma) q;  partials of |q|^2  (cgls.hpp:287-296)
template <class T> struct StepQ {
  static constexpr bool kSkipWhenDone = true, kTwoRanges = false;
  static constexpr int kRegion = kRegionQ, kRegion2 = -1;
  T* q; const T* sigma;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t& sa, dd_t&) const {
    T qv[V], dv[V];
    ldv<T, V>(q + i, qv); ldv<T, V>(sigma + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      qv[j] = (T)1 * t_sqrt(dv[j]) * qv[j];
      dd_acc(sa, (double)qv[j] * (double)qv[j]);
    }
    stv<T, V>(q + i, qv);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// STEP_XR:  (n) x = alpha p + x;  s = (-shift / (1 sqrt(Tau))) x;  partials of |x|^2
//           (m) r = -alpha q + r;  t = sqrt(Sigma) r      (cgls.hpp:311-325, :352-354)
template <class T> struct StepXR {
  static constexpr bool kSkipWhenDone = true, kTwoRanges = true;
  static constexpr int kRegion = kRegionX, kRegion2 = -1;
  T* x; const T* p; T* s; const T* tau; T* r; const T* q; const T* sigma; T* t; T negshift;     // s == nullptr: the consumer forms it from x
  T alpha, neg_alpha;
  __device__ void load(const CgState* st) { alpha = (T)st->alpha; neg_alpha = (T)st->neg_alpha; }
  template <int V> __device__ void range0(size_t i, dd_t& sa, dd_t&) const {
    T xv[V], pv[V], dv[V], sv[V];
    ldv<T, V>(x + i, xv); ldv<T, V>(p + i, pv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      xv[j] = alpha * pv[j] + xv[j];
      sv[j] = (negshift / ((T)1 * t_sqrt(dv[j]))) * xv[j];
      dd_acc(sa, (double)xv[j] * (double)xv[j]);
    }
    stv<T, V>(x + i, xv);
    if (s) stv<T, V>(s + i, sv);
  }
  template <int V> __device__ void range1(size_t i, dd_t&, dd_t&) const {
    T rv[V], qv[V], dv[V], tv[V];
    ldv<T, V>(r + i, rv); ldv<T, V>(q + i, qv); ldv<T, V>(sigma + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      rv[j] = neg_alpha * qv[j] + rv[j];
      tv[j] = t_sqrt(dv[j]) * rv[j];
    }
    stv<T, V>(r + i, rv); stv<T, V>(t + i, tv);
  }
};
// STEP_S (n), after s += K^T t:  s = 1 sqrt(Tau) s;  partials of |s|^2  (cgls.hpp:326-340)
template <class T> struct StepS {
  static constexpr bool kSkipWhenDone = true, kTwoRanges = false;
  static constexpr int kRegion = kRegionS, kRegion2 = -1;
  T* s; const T* tau;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t& sa, dd_t&) const {
    T sv[V], dv[V];
    ldv<T, V>(s + i, sv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      sv[j] = (T)1 * t_sqrt(dv[j]) * sv[j];
      dd_acc(sa, (double)sv[j] * (double)sv[j]);
    }
    stv<T, V>(s + i, sv);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// STEP_P (n):  p = beta p + s;  t = sqrt(Tau) p [gemv1 of the next q = A p];  partials of |p|^2  (cgls.hpp:341-351, :287-296)
template <class T> struct StepP {
  static constexpr bool kSkipWhenDone = true, kTwoRanges = false;
  static constexpr int kRegion = kRegionP, kRegion2 = -1;
  T* p; const T* s; T* t; const T* tau; T beta;
  __device__ void load(const CgState* st) { beta = (T)st->beta; }
  template <int V> __device__ void range0(size_t i, dd_t& sa, dd_t&) const {
    T pv[V], sv[V], dv[V], tv[V];
    ldv<T, V>(p + i, pv); ldv<T, V>(s + i, sv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      pv[j] = beta * pv[j] + sv[j];
      tv[j] = t_sqrt(dv[j]) * pv[j];
      dd_acc(sa, (double)pv[j] * (double)pv[j]);
    }
    stv<T, V>(p + i, pv); stv<T, V>(t + i, tv);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};

// ---- ADMM outer iteration (BackendADMM::PerformIteration, backend_admm.cu:355-665) ---------------
// The reference runs one thrust::transform per functor plus device-to-device copies between them;
// the stages below group the functors that touch the same element into one pass each.  Operand
// names are the reference's members.
//
// PRE_X (n):  temp1 = (alpha x_half + (1 - alpha) x_proj + x_dual) / sqrt(Tau)   temp1_functor :53-67
//             x_proj = temp3 (CG warm start :393);  temp3 = sqrt(Tau) temp1       gemv_functor1 of :399
template <class T> struct AdmmPreX {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* x_half; T* x_proj; const T* x_dual; const T* tau; T* temp1; T* temp3; T alpha;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t&, dd_t&) const {
    T a[V], b[V], c[V], d[V], w[V], o[V], t[V];
    ldv<T, V>(x_half + i, a); ldv<T, V>(x_proj + i, b); ldv<T, V>(x_dual + i, c); ldv<T, V>(tau + i, d); ldv<T, V>(temp3 + i, w);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      o[j] = (alpha * a[j] + (1 - alpha) * b[j] + c[j]) / sq;
      t[j] = sq * o[j];
    }
    stv<T, V>(temp1 + i, o); stv<T, V>(x_proj + i, w); stv<T, V>(temp3 + i, t);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// PRE_Z (m):  temp2 = sqrt(Sigma) (z_half + z_dual)                              temp2_functor :70-81
//             z_dual = (1 / (-1 sqrt(Sigma))) temp2                              gemv_functor2 of z_dual = temp2 - A temp1 (:399)
template <class T> struct AdmmPreZ {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* z_half; T* z_dual; const T* sigma; T* temp2;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t&, dd_t&) const {
    T a[V], b[V], d[V], o[V], z[V];
    ldv<T, V>(z_half + i, a); ldv<T, V>(z_dual + i, b); ldv<T, V>(sigma + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      o[j] = sq * (a[j] + b[j]);
      z[j] = ((T)1 / ((T)-1 * sq)) * o[j];
    }
    stv<T, V>(temp2 + i, o); stv<T, V>(z_dual + i, z);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// PRE_Z2 (m), after z_dual += K temp3:  z_dual = -1 sqrt(Sigma) z_dual            gemv_functor3 of :399
template <class T> struct AdmmPreZ2 {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  T* z_dual; const T* sigma;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t&, dd_t&) const {
    T z[V], d[V];
    ldv<T, V>(z_dual + i, z); ldv<T, V>(sigma + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) z[j] = (T)-1 * t_sqrt(d[j]) * z[j];
    stv<T, V>(z_dual + i, z);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// POST_X (n), after the CG solve:  temp3 = x_proj (:437);  x_proj = sqrt(Tau) (x_proj + temp1)   x_proj_functor :447-456
template <class T> struct AdmmPostX {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  T* x_proj; const T* temp1; const T* tau; T* temp3;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t&, dd_t&) const {
    T x[V], a[V], d[V], o[V];
    ldv<T, V>(x_proj + i, x); ldv<T, V>(temp1 + i, a); ldv<T, V>(tau + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) o[j] = t_sqrt(d[j]) * (x[j] + a[j]);
    stv<T, V>(temp3 + i, x); stv<T, V>(x_proj + i, o);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// POST_XZ, after z_proj = K x_proj:
//   (n) x_dual = temp1 sqrt(Tau) - x_proj  x_dual_functor :464-477;  temp1 = x_proj - x_dual  (prox_g argument :499)
//   (m) z_dual = temp2 / sqrt(Sigma) - z_proj  z_dual_functor :480-493;  temp2 = z_proj - z_dual  (prox_f argument :514)
template <class T> struct AdmmPostXZ {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = true;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* x_proj; T* x_dual; T* temp1; const T* tau; const T* z_proj; T* z_dual; T* temp2; const T* sigma;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t&, dd_t&) const {
    T a[V], b[V], d[V], u[V], o[V];
    ldv<T, V>(temp1 + i, a); ldv<T, V>(x_proj + i, b); ldv<T, V>(tau + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      u[j] = a[j] * t_sqrt(d[j]) - b[j];
      o[j] = b[j] - u[j];
    }
    stv<T, V>(x_dual + i, u); stv<T, V>(temp1 + i, o);
  }
  template <int V> __device__ void range1(size_t i, dd_t&, dd_t&) const {
    T a[V], b[V], d[V], u[V], o[V];
    ldv<T, V>(temp2 + i, a); ldv<T, V>(z_proj + i, b); ldv<T, V>(sigma + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      u[j] = a[j] / t_sqrt(d[j]) - b[j];
      o[j] = b[j] - u[j];
    }
    stv<T, V>(z_dual + i, u); stv<T, V>(temp2 + i, o);
  }
};
// get_dual_functor (backend_admm.cu:181-196): -rho scaling^expo (half - proj + dual); the reference calls it
// with expo = +1 / -1 only, for which pow() is the identity / the reciprocal
template <class T> __device__ __forceinline__ T get_dual(T rho, T scal, T expo, T half, T proj, T dual) {
  const T pw = expo == (T)1 ? scal : expo == (T)-1 ? (T)1 / scal : t_pow(scal, expo);
  return -rho * pw * (half - proj + dual);
}
// RES_Z (m), with kx = K x_half:  primal residual |sqrt(Sigma) (kx - z_half)|, |sqrt(Sigma) z_half|  (:541-566);
//             kx := y = get_dual(z_half, z_proj, z_dual, Sigma, +1)  (the K^T y of the dual residual, :596-606)
template <class T> struct AdmmResZ {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = kRegionQ, kRegion2 = -1;
  T* kx; const T* z_half; const T* z_proj; const T* z_dual; const T* sigma; T rho;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t& sa, dd_t& sb) const {
    T k[V], h[V], pj[V], du[V], d[V], y[V];
    ldv<T, V>(kx + i, k); ldv<T, V>(z_half + i, h); ldv<T, V>(z_proj + i, pj); ldv<T, V>(z_dual + i, du); ldv<T, V>(sigma + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      const T pr = sq * ((T)1.0 * k[j] + (T)-1.0 * h[j]);      // temp2 = -z_half; temp2 += K x_half; scaled
      const T pv = sq * h[j];
      dd_acc(sa, (double)pr * (double)pr);
      dd_acc(sb, (double)pv * (double)pv);
      y[j] = get_dual<T>(rho, d[j], (T)1, h[j], pj[j], du[j]);
    }
    stv<T, V>(kx + i, y);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// RES_X (n), with kty = K^T y:  w = get_dual(x_half, x_proj, x_dual, Tau, -1);  |sqrt(Tau) w| (dual variable norm, :570-590)
//             and |sqrt(Tau) (w + kty)| (dual residual, :596-616)
template <class T> struct AdmmResX {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = kRegionP, kRegion2 = -1;
  const T* kty; const T* x_half; const T* x_proj; const T* x_dual; const T* tau; T rho;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t& sa, dd_t& sb) const {
    T k[V], h[V], pj[V], du[V], d[V];
    ldv<T, V>(kty + i, k); ldv<T, V>(x_half + i, h); ldv<T, V>(x_proj + i, pj); ldv<T, V>(x_dual + i, du); ldv<T, V>(tau + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      const T w = get_dual<T>(rho, d[j], (T)-1, h[j], pj[j], du[j]);
      const T dv = sq * w;
      const T dr = sq * ((T)1.0 * k[j] + w);
      dd_acc(sa, (double)dr * (double)dr);
      dd_acc(sb, (double)dv * (double)dv);
    }
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};

// out4 = {sqrt(sum RES_Z a), sqrt(sum RES_Z b), sqrt(sum RES_X a), sqrt(sum RES_X b)}
//      = {primal residual, primal variable norm, dual residual, dual variable norm}; out4 may be pinned host memory
__global__ void __launch_bounds__(kBlock) admm_residual_fold_kernel(double* out4, const double* ws, unsigned gz, unsigned gx) {
  const double* rz = region(const_cast<double*>(ws), kRegionQ);
  const double* rx = region(const_cast<double*>(ws), kRegionP);
  double a, b, c, d;                        // (two folds at a time: half the barriers of four separate ones)
  fold_dd2(rz, gz, 4, rz + 2, gz, 4, a, b);
  fold_dd2(rx, gx, 4, rx + 2, gx, 4, c, d);
  if (threadIdx.x == 0) { out4[0] = sqrt(a); out4[1] = sqrt(b); out4[2] = sqrt(c); out4[3] = sqrt(d); }
}

// ---- normest power iteration (Problem::normest, problem.cu:429-500) --------------------------------
// |Sigma^(1/2) K Tau^(1/2)| by power iteration: per round the reference runs four scaling passes, two nrm2 and a divide
// around K and K^T; fused here into three passes (same expressions, same roundings):
// NORMEST_A (n):  x_temp = sqrt(Tau) (x / norm_x)     [the divide of the previous round folded in; norm_x = 0: first round, no divide]
template <class T> struct NormestA {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* x; const T* tau; T* x_temp; T norm_x; bool divide; const double* norm_from;
  __device__ void load(const CgState*) { if (norm_from) { norm_x = (T)*norm_from; divide = *norm_from != 0.0; } }
  template <int V> __device__ void range0(size_t i, dd_t&, dd_t&) const {
    T xv[V], dv[V], o[V];
    ldv<T, V>(x + i, xv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) { const T t = divide ? xv[j] / norm_x : xv[j]; o[j] = t_sqrt(dv[j]) * t; }
    stv<T, V>(x_temp + i, o);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// NORMEST_B (m), after ax = K x_temp:  a = sqrt(Sigma) ax;  partials of |a|^2;  ax = sqrt(Sigma) a
template <class T> struct NormestB {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = kRegionQ, kRegion2 = -1;
  T* ax; const T* sigma;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t& sa, dd_t&) const {
    T v[V], dv[V];
    ldv<T, V>(ax + i, v); ldv<T, V>(sigma + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(dv[j]);
      const T a = sq * v[j];
      dd_acc(sa, (double)a * (double)a);
      v[j] = sq * a;
    }
    stv<T, V>(ax + i, v);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// NORMEST_C (n), after x_temp = K^T ax:  x = sqrt(Tau) x_temp;  partials of |x|^2
template <class T> struct NormestC {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = kRegionP, kRegion2 = -1;
  T* x; const T* x_temp; const T* tau;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t& sa, dd_t&) const {
    T v[V], dv[V];
    ldv<T, V>(x_temp + i, v); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) { v[j] = t_sqrt(dv[j]) * v[j]; dd_acc(sa, (double)v[j] * (double)v[j]); }
    stv<T, V>(x + i, v);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// out[0] = sqrt(sum of the first components of g partial pairs of region r); out may be pinned host memory
__global__ void __launch_bounds__(kBlock) sqrt_fold_kernel(double* out, const double* ws, int r, unsigned g) {
  const double a = fold_dd(region(const_cast<double*>(ws), r), g, 4);
  if (threadIdx.x == 0) out[0] = sqrt(a);
}

template <class T>
static unsigned stage_grid(size_t nmax, bool vec) {
  unsigned g = grid_for(vec ? nmax / VecOf<T>::N : nmax, 2);
  return g > (unsigned)kReduceBlocks ? (unsigned)kReduceBlocks : g;
}

template <class T, class F>
static int launch_stage(const char* name, F f, size_t n0, size_t n1, bool vec, const prost_hip_cgls_desc* d, hipStream_t st) {
  constexpr int V = VecOf<T>::N;
  const unsigned g = stage_grid<T>(n0 > n1 ? n0 : n1, vec);
  const CgState* state = static_cast<const CgState*>(d->state);
  double* ws = static_cast<double*>(d->workspace);
  if (vec) hipLaunchKernelGGL((cg_stage_kernel<T, V, F>), dim3(g), dim3(kBlock), 0, st, f, n0, n1, state, ws);
  else hipLaunchKernelGGL((cg_stage_kernel<T, 1, F>), dim3(g), dim3(kBlock), 0, st, f, n0, n1, state, ws);
  PH_LAUNCH_END(name);
}

template <class T, int WHICH>
static int launch_scalars(const prost_hip_cgls_desc* d, unsigned g0, unsigned g1, hipStream_t st, unsigned stride = 4) {
  ScalarArgs a{d->shift, d->tol, (double)std::numeric_limits<T>::epsilon(), g0, g1, d->host_done, d->epoch, stride};
  hipLaunchKernelGGL((cg_scalar_kernel<T, WHICH>), dim3(1), dim3(kBlock), 0, st, static_cast<CgState*>(d->state),
                     static_cast<const double*>(d->workspace), a);
  PH_LAUNCH_END("cgls scalars");
}

template <class T>
static int cgls_stage(int stage, const prost_hip_cgls_desc* d, void* stream) {
  if (!d || !d->state || !d->workspace) { set_error("cgls_stage: state and workspace are required"); return 1; }
  hipStream_t st = as_stream(stream);
  const T* b = static_cast<const T*>(d->b);
  T* x = static_cast<T*>(d->x); T* p = static_cast<T*>(d->p); T* q = static_cast<T*>(d->q);
  T* r = static_cast<T*>(d->r); T* s = static_cast<T*>(d->s); T* t = static_cast<T*>(d->t);
  const T* sigma = static_cast<const T*>(d->sigma); const T* tau = static_cast<const T*>(d->tau);
  const size_t m = d->m, n = d->n;
  const T negshift = (T)(-d->shift);
  const bool vn = aligned16(x) && aligned16(p) && aligned16(s) && aligned16(t) && aligned16(tau) && n >= (size_t)VecOf<T>::N;
  const bool vm = aligned16(b) && aligned16(q) && aligned16(r) && aligned16(t) && aligned16(sigma) && m >= (size_t)VecOf<T>::N;
  const unsigned gn = stage_grid<T>(n, vn), gm = stage_grid<T>(m, vm), gx = stage_grid<T>(n > m ? n : m, vn && vm);
  int rc;
  switch (stage) {
    case PROST_CGLS_INIT_X:
      if ((rc = launch_stage<T>("cgls init_x", InitX<T>{x, tau, t, s, negshift}, n, 0, vn, d, st))) return rc;
      return launch_scalars<T, kScalarsInitX>(d, gn, 0, st);
    case PROST_CGLS_INIT_R: return launch_stage<T>("cgls init_r", InitR<T>{b, sigma, r}, m, 0, vm, d, st);
    case PROST_CGLS_INIT_R2: return launch_stage<T>("cgls init_r2", InitR2<T>{b, sigma, r, t, false}, m, 0, vm, d, st);
    case PROST_CGLS_INIT_S:
      if ((rc = launch_stage<T>("cgls init_s", InitS<T>{s, p, t, tau}, n, 0, vn, d, st))) return rc;
      return launch_scalars<T, kScalarsInitS>(d, gn, 0, st);
    case PROST_CGLS_STEP_Q:
      if ((rc = launch_stage<T>("cgls step_q", StepQ<T>{q, sigma}, m, 0, vm, d, st))) return rc;
      return launch_scalars<T, kScalarsAlpha>(d, gm, gn, st);
    case PROST_CGLS_STEP_XR:
      return launch_stage<T>("cgls step_xr", StepXR<T>{x, p, s, tau, r, q, sigma, t, negshift, (T)0, (T)0}, n, m, vn && vm, d, st);
    case PROST_CGLS_STEP_S:
      if ((rc = launch_stage<T>("cgls step_s", StepS<T>{s, tau}, n, 0, vn, d, st))) return rc;
      return launch_scalars<T, kScalarsBeta>(d, gn, gx, st);
    case PROST_CGLS_STEP_P: return launch_stage<T>("cgls step_p", StepP<T>{p, s, t, tau, (T)0}, n, 0, vn, d, st);
    default: set_error("cgls_stage: unknown stage"); return 1;
  }
}

template <class T>
static int admm_stage(int stage, const prost_hip_admm_desc* d, void* stream) {
  if (!d || !d->workspace) { set_error("admm_stage: workspace is required"); return 1; }
  hipStream_t st = as_stream(stream);
  T* x_half = static_cast<T*>(d->x_half); T* x_proj = static_cast<T*>(d->x_proj); T* x_dual = static_cast<T*>(d->x_dual);
  T* z_half = static_cast<T*>(d->z_half); T* z_proj = static_cast<T*>(d->z_proj); T* z_dual = static_cast<T*>(d->z_dual);
  T* temp1 = static_cast<T*>(d->temp1); T* temp2 = static_cast<T*>(d->temp2); T* temp3 = static_cast<T*>(d->temp3);
  T* kx = static_cast<T*>(d->kx); T* kty = static_cast<T*>(d->kty);
  const T* sigma = static_cast<const T*>(d->sigma); const T* tau = static_cast<const T*>(d->tau);
  const size_t m = d->m, n = d->n;
  bool vn = aligned16(x_half) && aligned16(x_proj) && aligned16(x_dual) && aligned16(temp1) && aligned16(temp3) && aligned16(tau) && aligned16(kty) &&
            n >= (size_t)VecOf<T>::N;
  bool vm = aligned16(z_half) && aligned16(z_proj) && aligned16(z_dual) && aligned16(temp2) && aligned16(sigma) && aligned16(kx) && m >= (size_t)VecOf<T>::N;
  prost_hip_cgls_desc c{};                   // launch_stage only reads state / workspace; these stages have no scalar record
  c.state = nullptr; c.workspace = d->workspace;
  int rc;
  switch (stage) {
    case PROST_ADMM_STAGE_PRE_X:
      return launch_stage<T>("admm pre_x", AdmmPreX<T>{x_half, x_proj, x_dual, tau, temp1, temp3, (T)d->alpha}, n, 0, vn, &c, st);
    case PROST_ADMM_STAGE_PRE_Z: return launch_stage<T>("admm pre_z", AdmmPreZ<T>{z_half, z_dual, sigma, temp2}, m, 0, vm, &c, st);
    case PROST_ADMM_STAGE_PRE_Z2: return launch_stage<T>("admm pre_z2", AdmmPreZ2<T>{z_dual, sigma}, m, 0, vm, &c, st);
    case PROST_ADMM_STAGE_POST_X: return launch_stage<T>("admm post_x", AdmmPostX<T>{x_proj, temp1, tau, temp3}, n, 0, vn, &c, st);
    case PROST_ADMM_STAGE_POST_XZ:
      return launch_stage<T>("admm post_xz", AdmmPostXZ<T>{x_proj, x_dual, temp1, tau, z_proj, z_dual, temp2, sigma}, n, m, vn && vm, &c, st);
    case PROST_ADMM_STAGE_RES_Z: return launch_stage<T>("admm res_z", AdmmResZ<T>{kx, z_half, z_proj, z_dual, sigma, (T)d->rho}, m, 0, vm, &c, st);
    case PROST_ADMM_STAGE_RES_X:
      if (!d->out4) { set_error("admm_stage: RES_X needs out4"); return 1; }
      if ((rc = launch_stage<T>("admm res_x", AdmmResX<T>{kty, x_half, x_proj, x_dual, tau, (T)d->rho}, n, 0, vn, &c, st))) return rc;
      hipLaunchKernelGGL(admm_residual_fold_kernel, dim3(1), dim3(kBlock), 0, st, d->out4, static_cast<const double*>(d->workspace),
                         stage_grid<T>(m, vm), stage_grid<T>(n, vn));
      PH_LAUNCH_END("admm residual fold");
    default: set_error("admm_stage: unknown stage"); return 1;
  }
}

template <class T>
static int normest_stage(int stage, const prost_hip_normest_desc* d, void* stream) {
  if (!d || !d->workspace) { set_error("normest_stage: workspace is required"); return 1; }
  hipStream_t st = as_stream(stream);
  T* x = static_cast<T*>(d->x); T* x_temp = static_cast<T*>(d->x_temp); T* ax = static_cast<T*>(d->ax);
  const T* sigma = static_cast<const T*>(d->sigma); const T* tau = static_cast<const T*>(d->tau);
  const size_t m = d->m, n = d->n;
  const bool vn = aligned16(x) && aligned16(x_temp) && aligned16(tau) && n >= (size_t)VecOf<T>::N;
  const bool vm = aligned16(ax) && aligned16(sigma) && m >= (size_t)VecOf<T>::N;
  prost_hip_cgls_desc c{};
  c.state = nullptr; c.workspace = d->workspace;
  int rc;
  switch (stage) {
    case PROST_NORMEST_A:
      return launch_stage<T>("normest a", NormestA<T>{x, tau, x_temp, (T)d->norm_x, d->norm_x != 0.0, d->norm_x_from}, n, 0, vn, &c, st);
    case PROST_NORMEST_B:
      if (!d->out) { set_error("normest_stage: out is required"); return 1; }
      if ((rc = launch_stage<T>("normest b", NormestB<T>{ax, sigma}, m, 0, vm, &c, st))) return rc;
      hipLaunchKernelGGL(sqrt_fold_kernel, dim3(1), dim3(kBlock), 0, st, d->out, static_cast<const double*>(d->workspace), (int)kRegionQ, stage_grid<T>(m, vm));
      PH_LAUNCH_END("normest fold");
    case PROST_NORMEST_C:
      if (!d->out) { set_error("normest_stage: out is required"); return 1; }
      if ((rc = launch_stage<T>("normest c", NormestC<T>{x, x_temp, tau}, n, 0, vn, &c, st))) return rc;
      hipLaunchKernelGGL(sqrt_fold_kernel, dim3(1), dim3(kBlock), 0, st, d->out + 1, static_cast<const double*>(d->workspace), (int)kRegionP, stage_grid<T>(n, vn));
      PH_LAUNCH_END("normest fold");
    default: set_error("normest_stage: unknown stage"); return 1;
  }
}


// ---- fused operator: CG rounds in FOUR launches, ADMM outer stages with the operator inside -------------------
// Kernel durations, not launch gaps, are what the staged sequence above is made of at the C4 size (n = 2 Mi, m = 5 Mi:
// profiles/r01_c4_admm_kernel_stats.csv -- per CG round 90 us in ten kernels, the device never idle): STEP_XR moves its
// 140 MB at 6.1 TB/s, but the operator is four small kernels at ~3.2 TB/s, each product is written unscaled and re-read by
// the stage that scales it, and two one-workgroup scalar kernels cost 5.8 us each.  Here
//   * the operator is applied by ONE kernel per direction whatever its blocks are: the thread that owns VEC consecutive
//     output elements evaluates their CSR rows / gradient stencil rows for every block that covers them, in block order,
//     with the expressions of csr_spmv_kernel<., 1, .> / grad_fwd_kernel / grad_adj_kernel (LinearOperator::Eval: zero
//     fill, then EvalAdd per block), and the stage that follows the product is its EPILOGUE -- the product never reaches
//     memory unscaled;
//   * the scalar kernels are gone: every workgroup of the CONSUMING kernel folds the partial sums itself (at most 2048
//     values, identical order in every workgroup, hence identical results) and forms alpha / beta / the stopping test in
//     double exactly as cg_scalar_kernel does; workgroup 0 records the result;
//   * the scalar record is an ARRAY: round j reads record j and writes record j + 1, so no kernel reads a field that
//     another workgroup of the same launch writes.
// Round:  OP_FWD<FwdQ>  |  STEP_XR2 (alpha)  |  OP_ADJ<AdjS>  |  STEP_P2 (beta, stopping test)
// out = E(K in) / out = E(v0 + K^T in) with the stage functor E as epilogue; `count` output elements; E::kRegion >= 0: one
// partial (pair) per workgroup.  E::prologue runs in every thread of every workgroup before the loop (block-wide folds).
// A wavefront takes 64 VEC consecutive elements per step, lane l the VEC elements from (w0 + l) VEC; the last, partly filled
// step of the range and the count % VEC tail run one element per lane.
template <class T, int VEC, bool ADJ, class E>
__global__ void __launch_bounds__(kBlock) op_stage_kernel(const FusedOpDev* __restrict__ opp, E e, const T* __restrict__ in, size_t count, const CgState* cur, double* ws) {
  const PROST_CONSTANT FusedOpDev& op = *as_constant(opp);
  if (E::kSkipWhenDone && cur->done) return;
  e.prologue(cur, ws);
  dd_t sa{0.0, 0.0}, sb{0.0, 0.0};
  const unsigned lane = threadIdx.x & (kWave - 1);
  const size_t nv = VEC > 1 ? (count / ((size_t)VEC * kWave)) * kWave : count;          // vector groups in full wavefront steps
  for (size_t i0 = (size_t)blockIdx.x * kBlock + (threadIdx.x - lane); i0 < nv; i0 += (size_t)gridDim.x * kBlock) {
    const size_t i = (i0 + lane) * VEC;
    T kv[VEC];
    if (VEC > 1 || i < count) {
      if constexpr (ADJ) { e.template init<VEC>(i, kv); op_adj_cols<T, VEC>(op, i, i0 * VEC, in, kv); }
      else if constexpr (E::kFwdInit) { e.template init<VEC>(i, kv); op_fwd_rows<T, VEC, false>(op, i, i0 * VEC, in, kv); }
      else op_fwd_rows<T, VEC>(op, i, i0 * VEC, in, kv);
      e.template apply<VEC>(i, kv, sa, sb);
    }
  }
  if (VEC > 1) {
    for (size_t i = nv * VEC + (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock) {
      T kv[1];
      if constexpr (ADJ) { e.template init<1>(i, kv); op_adj_cols<T, 1>(op, i, i, in, kv); }
      else if constexpr (E::kFwdInit) { e.template init<1>(i, kv); op_fwd_rows<T, 1, false>(op, i, i, in, kv); }
      else op_fwd_rows<T, 1>(op, i, i, in, kv);
      e.template apply<1>(i, kv, sa, sb);
    }
  }
  if (E::kRegion >= 0) {
    if (E::kPairs) {
      block_dd_store2(sa, sb, region(ws, E::kRegion), blockIdx.x);
    } else {
      block_dd_store1(sa, region(ws, E::kRegion), blockIdx.x);
      if (E::kRegion2 >= 0 && threadIdx.x == 0) {
        region(ws, E::kRegion2)[2 * blockIdx.x] = region(ws, E::kRegion)[2 * blockIdx.x];
        region(ws, E::kRegion2)[2 * blockIdx.x + 1] = region(ws, E::kRegion)[2 * blockIdx.x + 1];
      }
    }
  }
}

// ---- epilogues (the stage structs above, fed with the product instead of reading it back) ----
// FWD_Q (m):  q = 1 sqrt(Sigma) (K t);  |q|^2                                   [K->Eval(q, t, 0) ; STEP_Q]
template <class T> struct EpiFwdQ {
  static constexpr bool kFwdInit = false;
  static constexpr bool kPairs = false;
  static constexpr bool kSkipWhenDone = true;
  static constexpr int kRegion = kRegionQ, kRegion2 = -1;
  T* q; const T* sigma;
  __device__ void prologue(const CgState*, double*) {}
  template <int V> __device__ void apply(size_t i, const T (&kv)[V], dd_t& sa, dd_t&) const {
    T dv[V], qv[V];
    ldv<T, V>(sigma + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) { qv[j] = (T)1 * t_sqrt(dv[j]) * kv[j]; dd_acc(sa, (double)qv[j] * (double)qv[j]); }
    stv<T, V>(q + i, qv);
  }
};
// ADJ_S (n):  s = 1 sqrt(Tau) (s0 + K^T t) with s0 = (-shift / (1 sqrt(Tau))) x formed here (STEP_XR's expression: the vector
//             STEP_XR would write and this kernel read back);  |s|^2            [K->EvalAdjoint(s, t, 1) ; STEP_S]
template <class T> struct EpiAdjS {
  static constexpr bool kFwdInit = false;
  static constexpr bool kPairs = false;
  static constexpr bool kSkipWhenDone = true;
  static constexpr int kRegion = kRegionS, kRegion2 = -1;
  T* s; const T* tau; const T* x; T negshift;
  __device__ void prologue(const CgState*, double*) {}
  template <int V> __device__ void init(size_t i, T (&v)[V]) const {
    T xv[V], dv[V];
    ldv<T, V>(x + i, xv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) v[j] = (negshift / ((T)1 * t_sqrt(dv[j]))) * xv[j];
  }
  template <int V> __device__ void apply(size_t i, const T (&kv)[V], dd_t& sa, dd_t&) const {
    T dv[V], sv[V];
    ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) { sv[j] = (T)1 * t_sqrt(dv[j]) * kv[j]; dd_acc(sa, (double)sv[j] * (double)sv[j]); }
    stv<T, V>(s + i, sv);
  }
};
// INIT_RK (m):  r = (1 / (-1 sqrt(Sigma))) b ; r += K t ; r = normx > 0 ? -1 sqrt(Sigma) r : b ; tm = sqrt(Sigma) r
//               [INIT_R ; K->Eval(r, t, 1) ; INIT_R2 -- tm goes to its own buffer: t is being read by other threads]
// (the product is ADDED to r block by block, in block order -- kFwdInit: the rows start from INIT_R's value, not from zero)
template <class T> struct EpiInitRK {
  static constexpr bool kFwdInit = true;
  static constexpr bool kPairs = false;
  static constexpr bool kSkipWhenDone = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* b; const T* sigma; T* r; T* tm; bool nonzero;
  __device__ void prologue(const CgState* st, double*) { nonzero = st->normx > 0.; }
  template <int V> __device__ void init(size_t i, T (&v)[V]) const {
    T bv[V], dv[V];
    ldv<T, V>(b + i, bv); ldv<T, V>(sigma + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) v[j] = ((T)1 / ((T)-1 * t_sqrt(dv[j]))) * bv[j];
  }
  template <int V> __device__ void apply(size_t i, const T (&kv)[V], dd_t&, dd_t&) const {
    T bv[V], dv[V], rv[V], tv[V];
    ldv<T, V>(b + i, bv); ldv<T, V>(sigma + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(dv[j]);
      T rr = kv[j];
      rr = nonzero ? (T)-1 * sq * rr : bv[j];
      rv[j] = rr;
      tv[j] = sq * rr;
    }
    stv<T, V>(r + i, rv); stv<T, V>(tm + i, tv);
  }
};
// INIT_SK (n):  s = 1 sqrt(Tau) (s + K^T tm) ; p = s ; t = sqrt(Tau) p ; |s|^2 = |p|^2       [K->EvalAdjoint(s, tm, 1) ; INIT_S]
template <class T> struct EpiInitSK {
  static constexpr bool kFwdInit = false;
  static constexpr bool kPairs = false;
  static constexpr bool kSkipWhenDone = false;
  static constexpr int kRegion = kRegionS, kRegion2 = kRegionP;
  T* s; T* p; T* t; const T* tau;
  __device__ void prologue(const CgState*, double*) {}
  template <int V> __device__ void init(size_t i, T (&v)[V]) const { ldv<T, V>(s + i, v); }
  template <int V> __device__ void apply(size_t i, const T (&kv)[V], dd_t& sa, dd_t&) const {
    T dv[V], sv[V], tv[V];
    ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(dv[j]);
      sv[j] = (T)1 * sq * kv[j];
      tv[j] = sq * sv[j];
      dd_acc(sa, (double)sv[j] * (double)sv[j]);
    }
    stv<T, V>(s + i, sv); stv<T, V>(p + i, sv); stv<T, V>(t + i, tv);
  }
};
// PRE_ZK (m):  temp2 = sqrt(Sigma) (z_half + z_dual) ; z_dual = (1 / (-1 sqrt(Sigma))) temp2 ; z_dual += K temp3 ;
//              z_dual = -1 sqrt(Sigma) z_dual                                    [PRE_Z ; K->Eval(z_dual, temp3, 1) ; PRE_Z2]
template <class T> struct EpiPreZK {
  static constexpr bool kFwdInit = true;             // the product is added to PRE_Z's z_dual block by block
  static constexpr bool kPairs = false;
  static constexpr bool kSkipWhenDone = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* z_half; T* z_dual; const T* sigma; T* temp2;
  __device__ void prologue(const CgState*, double*) {}
  template <int V> __device__ void init(size_t i, T (&v)[V]) const {
    T a[V], bb[V], d[V];
    ldv<T, V>(z_half + i, a); ldv<T, V>(z_dual + i, bb); ldv<T, V>(sigma + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      v[j] = ((T)1 / ((T)-1 * sq)) * (sq * (a[j] + bb[j]));
    }
  }
  template <int V> __device__ void apply(size_t i, const T (&kv)[V], dd_t&, dd_t&) const {
    T a[V], bb[V], d[V], o[V], z[V];
    ldv<T, V>(z_half + i, a); ldv<T, V>(z_dual + i, bb); ldv<T, V>(sigma + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      o[j] = sq * (a[j] + bb[j]);
      z[j] = (T)-1 * sq * kv[j];
    }
    stv<T, V>(temp2 + i, o); stv<T, V>(z_dual + i, z);
  }
};
// POST_ZK (m):  z_proj = K x_proj ; z_dual = temp2 / sqrt(Sigma) - z_proj ; temp2 = z_proj - z_dual
//               [K->Eval(z_proj, x_proj) ; the m half of POST_XZ]
template <class T> struct EpiPostZK {
  static constexpr bool kFwdInit = false;
  static constexpr bool kPairs = false;
  static constexpr bool kSkipWhenDone = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  T* z_proj; T* z_dual; T* temp2; const T* sigma;
  __device__ void prologue(const CgState*, double*) {}
  template <int V> __device__ void apply(size_t i, const T (&kv)[V], dd_t&, dd_t&) const {
    T a[V], d[V], u[V], o[V];
    ldv<T, V>(temp2 + i, a); ldv<T, V>(sigma + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      u[j] = a[j] / t_sqrt(d[j]) - kv[j];
      o[j] = kv[j] - u[j];
    }
    stv<T, V>(z_proj + i, kv); stv<T, V>(z_dual + i, u); stv<T, V>(temp2 + i, o);
  }
};
// POST_X2 (n, elementwise):  temp3 = x_proj ; x_proj = sqrt(Tau) (x_proj + temp1) ; x_dual = temp1 sqrt(Tau) - x_proj ;
//               temp1 = x_proj - x_dual                                          [POST_X ; the n half of POST_XZ]
template <class T> struct AdmmPostX2 {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  T* x_proj; T* temp1; const T* tau; T* temp3; T* x_dual;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, dd_t&, dd_t&) const {
    T x[V], a[V], d[V], o[V], u[V], w[V];
    ldv<T, V>(x_proj + i, x); ldv<T, V>(temp1 + i, a); ldv<T, V>(tau + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      o[j] = sq * (x[j] + a[j]);
      u[j] = a[j] * sq - o[j];
      w[j] = o[j] - u[j];
    }
    stv<T, V>(temp3 + i, x); stv<T, V>(x_proj + i, o); stv<T, V>(x_dual + i, u); stv<T, V>(temp1 + i, w);
  }
  template <int V> __device__ void range1(size_t, dd_t&, dd_t&) const {}
};
// RES_ZK (m):  kx = K x_half ; RES_Z (primal residual / variable norm sums; y = get_dual(...) stored in kx)
template <class T> struct EpiResZK {
  static constexpr bool kFwdInit = false;
  static constexpr bool kPairs = true;
  static constexpr bool kSkipWhenDone = false;
  static constexpr int kRegion = kRegionQ, kRegion2 = -1;
  T* kx; const T* z_half; const T* z_proj; const T* z_dual; const T* sigma; T rho;
  __device__ void prologue(const CgState*, double*) {}
  template <int V> __device__ void apply(size_t i, const T (&kv)[V], dd_t& sa, dd_t& sb) const {
    T h[V], pj[V], du[V], d[V], y[V];
    ldv<T, V>(z_half + i, h); ldv<T, V>(z_proj + i, pj); ldv<T, V>(z_dual + i, du); ldv<T, V>(sigma + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      const T pr = sq * ((T)1.0 * kv[j] + (T)-1.0 * h[j]);
      const T pv = sq * h[j];
      dd_acc(sa, (double)pr * (double)pr);
      dd_acc(sb, (double)pv * (double)pv);
      y[j] = get_dual<T>(rho, d[j], (T)1, h[j], pj[j], du[j]);
    }
    stv<T, V>(kx + i, y);
  }
};
// RES_XK (n):  kty = K^T y (not stored) ; RES_X (dual residual / variable norm sums)
template <class T> struct EpiResXK {
  static constexpr bool kFwdInit = false;
  static constexpr bool kPairs = true;
  static constexpr bool kSkipWhenDone = false;
  static constexpr int kRegion = kRegionP, kRegion2 = -1;
  const T* x_half; const T* x_proj; const T* x_dual; const T* tau; T rho;
  __device__ void prologue(const CgState*, double*) {}
  template <int V> __device__ void init(size_t, T (&v)[V]) const {
#pragma unroll
    for (int j = 0; j < V; j++) v[j] = 0;
  }
  template <int V> __device__ void apply(size_t i, const T (&kv)[V], dd_t& sa, dd_t& sb) const {
    T h[V], pj[V], du[V], d[V];
    ldv<T, V>(x_half + i, h); ldv<T, V>(x_proj + i, pj); ldv<T, V>(x_dual + i, du); ldv<T, V>(tau + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      const T w = get_dual<T>(rho, d[j], (T)-1, h[j], pj[j], du[j]);
      const T dv = sq * w;
      const T dr = sq * ((T)1.0 * kv[j] + w);
      dd_acc(sa, (double)dr * (double)dr);
      dd_acc(sb, (double)dv * (double)dv);
    }
  }
};

struct RoundScalars { double shift, eps; unsigned g_a, g_b; int* host_done; };

// STEP_XR with alpha formed here from the partials of |q|^2 (g_a pairs) and |p|^2 (g_b pairs)   (cgls.hpp:297-325)
template <class T, int VEC>
__global__ void __launch_bounds__(kBlock) cg_step_xr2_kernel(StepXR<T> f, size_t n, size_t m, const CgState* cur, CgState* nxt, double* ws, RoundScalars a) {
  if (cur->done) return;
  // operands of the first element group requested before the fold (see cg_step_p2_kernel)
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x, stride = (size_t)gridDim.x * kBlock;
  const size_t nvn = n / VEC;
  T xv[VEC], pv[VEC], dv[VEC];
  const bool first = tid < nvn;
  if (first) { ldv<T, VEC>(f.x + tid * VEC, xv); ldv<T, VEC>(f.p + tid * VEC, pv); ldv<T, VEC>(f.tau + tid * VEC, dv); }
  double s0, s1;
  fold_dd2(region(ws, kRegionQ), a.g_a, 2, region(ws, kRegionP), a.g_b, 2, s0, s1);
  const double normq = sqrt(s0), normp = sqrt(s1);
  double dlt = normq * normq + a.shift * normp * normp;
  const int indefinite = dlt <= 0. ? 1 : 0;
  if (dlt == 0.) dlt = a.eps;
  f.alpha = (T)(cur->gamma / dlt);
  f.neg_alpha = (T)(-cur->gamma / dlt);
  if (blockIdx.x == 0 && threadIdx.x == 0) nxt->indefinite = cur->indefinite | indefinite;
  dd_t sa{0.0, 0.0}, sb{0.0, 0.0};
  if (first) {                                       // StepXR::range0 on the prefetched operands
    T sv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      xv[j] = f.alpha * pv[j] + xv[j];
      sv[j] = (f.negshift / ((T)1 * t_sqrt(dv[j]))) * xv[j];
      dd_acc(sa, (double)xv[j] * (double)xv[j]);
    }
    stv<T, VEC>(f.x + tid * VEC, xv);
    if (f.s) stv<T, VEC>(f.s + tid * VEC, sv);
  }
  for (size_t i = tid + stride; i < nvn; i += stride) f.template range0<VEC>(i * VEC, sa, sb);
  if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < n - nvn * VEC) f.template range0<1>(nvn * VEC + threadIdx.x, sa, sb);
  {
    const size_t nv = m / VEC;
    for (size_t i = tid; i < nv; i += stride) f.template range1<VEC>(i * VEC, sa, sb);
    if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < m - nv * VEC) f.template range1<1>(nv * VEC + threadIdx.x, sa, sb);
  }
  block_dd_store1(sa, region(ws, kRegionX), blockIdx.x);
}
// STEP_P with beta and the stopping test formed here from the partials of |s|^2 (g_a) and |x|^2 (g_b)   (cgls.hpp:326-360);
// workgroup 0 writes the next record.  A round that finds the solve finished only hands the record on.
template <class T, int VEC>
__global__ void __launch_bounds__(kBlock) cg_step_p2_kernel(StepP<T> f, size_t n, const CgState* cur, CgState* nxt, double* ws, RoundScalars a) {
  if (cur->done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *nxt = *cur;
    return;
  }
  // the operands of this thread's first element group are requested BEFORE the fold: they do not depend on beta, and the fold
  // (a cache latency + two barriers in every workgroup) would otherwise sit in front of the first byte this kernel streams
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x, stride = (size_t)gridDim.x * kBlock;
  const size_t nv = n / VEC;
  T pv[VEC], sv[VEC], dv[VEC];
  const bool first = tid < nv;
  if (first) { ldv<T, VEC>(f.p + tid * VEC, pv); ldv<T, VEC>(f.s + tid * VEC, sv); ldv<T, VEC>(f.tau + tid * VEC, dv); }
  double s0, s1;
  fold_dd2(region(ws, kRegionS), a.g_a, 2, region(ws, kRegionX), a.g_b, 2, s0, s1);
  const double norms = sqrt(s0), gamma = norms * norms, normx = sqrt(s1);
  f.beta = (T)(gamma / cur->gamma);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int indefinite = nxt->indefinite;          // written by this round's STEP_XR2
    CgState r = *cur;
    r.indefinite = indefinite;
    r.norms = norms; r.gamma = gamma; r.beta = (double)f.beta; r.normx = normx;
    r.xmax = cur->xmax > normx ? cur->xmax : normx;
    if ((norms <= cur->norms0 * cur->tol) || (normx * cur->tol >= 1.)) {
      r.done = 1;
      if (a.host_done) __hip_atomic_store(a.host_done, cur->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      r.k = cur->k + 1;
    }
    *nxt = r;
  }
  dd_t sa{0.0, 0.0}, sb{0.0, 0.0};
  if (first) {                                       // StepP::range0 on the prefetched operands
    T tv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      pv[j] = f.beta * pv[j] + sv[j];
      tv[j] = t_sqrt(dv[j]) * pv[j];
      dd_acc(sa, (double)pv[j] * (double)pv[j]);
    }
    stv<T, VEC>(f.p + tid * VEC, pv); stv<T, VEC>(f.t + tid * VEC, tv);
  }
  for (size_t i = tid + stride; i < nv; i += stride) f.template range0<VEC>(i * VEC, sa, sb);
  if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < n - nv * VEC) f.template range0<1>(nv * VEC + threadIdx.x, sa, sb);
  block_dd_store1(sa, region(ws, kRegionP), blockIdx.x);
}

// Workgroups of a kernel whose partial sums every workgroup of the NEXT kernel folds itself.  A fold costs one memory latency
// plus g / 256 loads per thread whatever g is (the partial array stays in L1 / L2), so the PRODUCING kernels may use as many
// workgroups as the workspace has slots: the operator kernels -- chains of dependent loads (row start -> value, index ->
// gathered operand) -- run one VEC-row group per thread; the streaming stages cap at kFoldBlocks workgroups.
}  // namespace prost_hip
#include <cstring>
#include <mutex>
#include <vector>
namespace prost_hip {
const FusedOpDev* device_op(const prost_hip_fused_op* op) {
  static std::mutex mu;
  struct Entry { FusedOpDev host; FusedOpDev* dev; int device; };
  static std::vector<Entry> cache;
  FusedOpDev h;
  std::memset(&h, 0, sizeof(h));                      // (padding bytes: entries are compared with memcmp)
  {
    const FusedOpDev made = make_op(op);
    h.nblocks = made.nblocks;
    for (int b = 0; b < made.nblocks; b++) {
      OpBlockDev& D = h.b[b]; const OpBlockDev& S = made.b[b];
      D.kind = S.kind; D.row = S.row; D.col = S.col; D.nrows = S.nrows; D.ncols = S.ncols; D.nx = S.nx; D.ny = S.ny; D.L = S.L;
      D.val = S.val; D.ptr = S.ptr; D.ind = S.ind; D.val_t = S.val_t; D.ptr_t = S.ptr_t; D.ind_t = S.ind_t;
      D.ids = S.ids; D.pptr = S.pptr; D.rel = S.rel; D.pval = S.pval; D.ids_t = S.ids_t; D.pptr_t = S.pptr_t; D.rel_t = S.rel_t; D.pval_t = S.pval_t;
      D.anchor = S.anchor; D.anchor_t = S.anchor_t;
    }
  }
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) { set_error("device_op: no current device"); return nullptr; }
  std::lock_guard<std::mutex> g(mu);
  for (const Entry& e : cache) if (e.device == device && std::memcmp(&e.host, &h, sizeof(h)) == 0) return e.dev;
  // Tables are never freed: the pointer is handed out after the mutex is released, so another thread (the ranks of a multi-rank test
  // live in one process) may be about to launch with any cached table, and a table may belong to another device than the current one.
  // They are ~2 KB each; a process that has seen 4096 distinct operators stops caching the oldest half of the list instead (the
  // tables themselves stay allocated).
  if (cache.size() >= 4096) cache.erase(cache.begin(), cache.begin() + 2048);
  FusedOpDev* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(FusedOpDev)) != hipSuccess || hipMemcpy(d, &h, sizeof(FusedOpDev), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("device_op: cannot upload the operator table");
    return nullptr;
  }
  cache.push_back(Entry{h, d, device});
  return d;
}
constexpr unsigned kFoldBlocks = 2048;
constexpr unsigned kOpBlocks = 2048;
static unsigned fold_grid(size_t elements, unsigned per_thread, unsigned cap = kFoldBlocks) {
  const size_t need = (elements / per_thread + kBlock - 1) / kBlock;
  return (unsigned)(need < 1 ? 1 : need > cap ? cap : need);
}
static unsigned op_grid(size_t elements, unsigned per_thread) { return fold_grid(elements, per_thread, (unsigned)kReduceBlocks); }

template <class T, bool ADJ, class E>
static void launch_op(const prost_hip_fused_op* op, const E& e, const T* in, size_t count, bool vec, unsigned grid, const CgState* cur, double* ws, hipStream_t st) {
  constexpr int V = VecOf<T>::N;
  const FusedOpDev* dev = device_op(op);
  if (!dev) return;
  if (vec) PH_LAUNCH((op_stage_kernel<T, V, ADJ, E>), dim3(grid), dim3(kBlock), 0, st, dev, e, in, count, cur, ws);
  else PH_LAUNCH((op_stage_kernel<T, 1, ADJ, E>), dim3(grid), dim3(kBlock), 0, st, dev, e, in, count, cur, ws);
}

template <class T>
struct CgPtrs {
  const T* b; T *x, *p, *q, *r, *s, *t; const T *sigma, *tau; size_t m, n; bool vn, vm, vop;
  CgPtrs(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op)
      : b(static_cast<const T*>(d->b)), x(static_cast<T*>(d->x)), p(static_cast<T*>(d->p)), q(static_cast<T*>(d->q)), r(static_cast<T*>(d->r)),
        s(static_cast<T*>(d->s)), t(static_cast<T*>(d->t)), sigma(static_cast<const T*>(d->sigma)), tau(static_cast<const T*>(d->tau)), m(d->m), n(d->n) {
    constexpr int V = VecOf<T>::N;
    vn = aligned16(x) && aligned16(p) && aligned16(s) && aligned16(t) && aligned16(tau) && n >= (size_t)V;
    vm = aligned16(b) && aligned16(q) && aligned16(r) && aligned16(t) && aligned16(sigma) && m >= (size_t)V;
    vop = vn && vm && fused_op_vec_ok(op, V);
  }
};

template <class T>
static int cgls_round(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, int round, void* stream, void* const* ev8 = nullptr) {
  if (!d || !d->state || !d->workspace) { set_error("cgls_round: state and workspace are required"); return 1; }
  if (round < 0) { set_error("cgls_round: negative round"); return 1; }
  if (!fused_op_ok(op, d->m, d->n)) { set_error("cgls_round: unsupported operator description (prost_hip_fused_op_supported)"); return 1; }
  if (!device_op(op)) return 1;                        // (uploads the block table on first sight; launch_op finds it in the cache)
  hipStream_t st = as_stream(stream);
  constexpr int V = VecOf<T>::N;
  const CgPtrs<T> c(d, op);
  const unsigned gq = fold_grid(c.m, c.vop ? V : 1, kOpBlocks), gs = fold_grid(c.n, c.vop ? V : 1, kOpBlocks);
  // (STEP_P2 moves a quarter of what STEP_XR2 moves: half as many workgroups fold the partial sums in front of it)
  const unsigned gx = fold_grid(c.n > c.m ? c.n : c.m, c.vn && c.vm ? V : 1), gpp = fold_grid(c.n, c.vn ? V : 1, kFoldBlocks / 2);
  const unsigned gp = round == 0 ? gs : gpp;          // |p|^2 of round 0 comes from INIT_SK (grid gs)
  const CgState* cur = static_cast<const CgState*>(d->state) + round;
  CgState* nxt = static_cast<CgState*>(d->state) + round + 1;
  double* ws = static_cast<double*>(d->workspace);
  const double eps = (double)std::numeric_limits<T>::epsilon();
  // ev8: kernel k of the round stamps ev8[2 k] / ev8[2 k + 1] with its own begin / end (PH_LAUNCH, common.hpp)
  auto mark = [&](int k) { if (ev8 && ev8[2 * k] && ev8[2 * k + 1]) { g_launch_ev_start = (hipEvent_t)ev8[2 * k]; g_launch_ev_stop = (hipEvent_t)ev8[2 * k + 1]; } };
  mark(0);
  launch_op<T, false>(op, EpiFwdQ<T>{c.q, c.sigma}, c.t, c.m, c.vop, gq, cur, ws, st);
  mark(1);
  const StepXR<T> fx{c.x, c.p, nullptr, c.tau, c.r, c.q, c.sigma, c.t, (T)(-d->shift), (T)0, (T)0};
  const RoundScalars ax{d->shift, eps, gq, gp, nullptr};
  if (c.vn && c.vm) PH_LAUNCH((cg_step_xr2_kernel<T, V>), dim3(gx), dim3(kBlock), 0, st, fx, c.n, c.m, cur, nxt, ws, ax);
  else PH_LAUNCH((cg_step_xr2_kernel<T, 1>), dim3(gx), dim3(kBlock), 0, st, fx, c.n, c.m, cur, nxt, ws, ax);
  mark(2);
  launch_op<T, true>(op, EpiAdjS<T>{c.s, c.tau, c.x, (T)(-d->shift)}, c.t, c.n, c.vop, gs, cur, ws, st);
  mark(3);
  const StepP<T> fp{c.p, c.s, c.t, c.tau, (T)0};
  const RoundScalars ap{d->shift, eps, gs, gx, d->host_done};
  if (c.vn) PH_LAUNCH((cg_step_p2_kernel<T, V>), dim3(gpp), dim3(kBlock), 0, st, fp, c.n, cur, nxt, ws, ap);
  else PH_LAUNCH((cg_step_p2_kernel<T, 1>), dim3(gpp), dim3(kBlock), 0, st, fp, c.n, cur, nxt, ws, ap);
  PH_LAUNCH_END("cgls round");
}

// INIT_X ; [INIT_R ; r += K t ; INIT_R2] ; [s += K^T tm ; INIT_S] with the two operator applications inside their stages;
// tm (= sqrt(Sigma) r) passes through q, which the first round overwrites
template <class T>
static int cgls_init_fused(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, void* stream) {
  if (!d || !d->state || !d->workspace) { set_error("cgls_init_fused: state and workspace are required"); return 1; }
  if (!fused_op_ok(op, d->m, d->n)) { set_error("cgls_init_fused: unsupported operator description (prost_hip_fused_op_supported)"); return 1; }
  if (!device_op(op)) return 1;                        // (uploads the block table on first sight; launch_op finds it in the cache)
  hipStream_t st = as_stream(stream);
  constexpr int V = VecOf<T>::N;
  const CgPtrs<T> c(d, op);
  int rc;
  if ((rc = cgls_stage<T>(PROST_CGLS_INIT_X, d, stream))) return rc;               // t = sqrt(Tau) x, s = -shift x / sqrt(Tau), record 0
  const CgState* cur = static_cast<const CgState*>(d->state);
  double* ws = static_cast<double*>(d->workspace);
  launch_op<T, false>(op, EpiInitRK<T>{c.b, c.sigma, c.r, c.q, false}, c.t, c.m, c.vop, op_grid(c.m, c.vop ? V : 1), cur, ws, st);
  const unsigned gs = fold_grid(c.n, c.vop ? V : 1, kOpBlocks);          // = the grid round 0 expects for |p|^2
  launch_op<T, true>(op, EpiInitSK<T>{c.s, c.p, c.t, c.tau}, c.q, c.n, c.vop, gs, cur, ws, st);
  return launch_scalars<T, kScalarsInitS>(d, gs, 0, st, 2);
}

// ADMM outer iteration with the operator inside the stages (the caller runs the CG solve and the proxes in between):
//   PRE : PRE_X ; [PRE_Z ; z_dual += K temp3 ; PRE_Z2]
//   POST: [POST_X ; n half of POST_XZ] ; [z_proj = K x_proj ; m half of POST_XZ]
//   RES : [kx = K x_half ; RES_Z] ; [K^T kx ; RES_X] ; fold -> out4
template <class T>
static int admm_fused_stage(int stage, const prost_hip_admm_desc* d, const prost_hip_fused_op* op, void* stream) {
  if (!d || !d->workspace) { set_error("admm_fused_stage: workspace is required"); return 1; }
  if (!fused_op_ok(op, d->m, d->n)) { set_error("admm_fused_stage: unsupported operator description (prost_hip_fused_op_supported)"); return 1; }
  if (!device_op(op)) return 1;                        // (uploads the block table on first sight; launch_op finds it in the cache)
  hipStream_t st = as_stream(stream);
  constexpr int V = VecOf<T>::N;
  T* x_half = static_cast<T*>(d->x_half); T* x_proj = static_cast<T*>(d->x_proj); T* x_dual = static_cast<T*>(d->x_dual);
  T* z_half = static_cast<T*>(d->z_half); T* z_proj = static_cast<T*>(d->z_proj); T* z_dual = static_cast<T*>(d->z_dual);
  T* temp1 = static_cast<T*>(d->temp1); T* temp2 = static_cast<T*>(d->temp2); T* temp3 = static_cast<T*>(d->temp3);
  T* kx = static_cast<T*>(d->kx);
  const T* sigma = static_cast<const T*>(d->sigma); const T* tau = static_cast<const T*>(d->tau);
  const size_t m = d->m, n = d->n;
  const bool vn = aligned16(x_half) && aligned16(x_proj) && aligned16(x_dual) && aligned16(temp1) && aligned16(temp3) && aligned16(tau) && n >= (size_t)V;
  const bool vm = aligned16(z_half) && aligned16(z_proj) && aligned16(z_dual) && aligned16(temp2) && aligned16(sigma) && aligned16(kx) && m >= (size_t)V;
  const bool vop = vn && vm && fused_op_vec_ok(op, V);
  double* ws = static_cast<double*>(d->workspace);
  prost_hip_cgls_desc c{};
  c.state = nullptr; c.workspace = d->workspace;
  int rc;
  switch (stage) {
    case PROST_ADMM_FUSED_PRE:
      if ((rc = launch_stage<T>("admm pre_x", AdmmPreX<T>{x_half, x_proj, x_dual, tau, temp1, temp3, (T)d->alpha}, n, 0, vn, &c, st))) return rc;
      launch_op<T, false>(op, EpiPreZK<T>{z_half, z_dual, sigma, temp2}, temp3, m, vop, grid_for(vop ? m / V : m), nullptr, ws, st);
      PH_LAUNCH_END("admm pre_zk");
    case PROST_ADMM_FUSED_POST:
      if ((rc = launch_stage<T>("admm post_x2", AdmmPostX2<T>{x_proj, temp1, tau, temp3, x_dual}, n, 0, vn, &c, st))) return rc;
      launch_op<T, false>(op, EpiPostZK<T>{z_proj, z_dual, temp2, sigma}, x_proj, m, vop, grid_for(vop ? m / V : m), nullptr, ws, st);
      PH_LAUNCH_END("admm post_zk");
    case PROST_ADMM_FUSED_RES: {
      if (!d->out4) { set_error("admm_fused_stage: RES needs out4"); return 1; }
      const unsigned gz = fold_grid(m, vop ? V : 1, 2048), gx = fold_grid(n, vop ? V : 1, 2048);
      launch_op<T, false>(op, EpiResZK<T>{kx, z_half, z_proj, z_dual, sigma, (T)d->rho}, x_half, m, vop, gz, nullptr, ws, st);
      launch_op<T, true>(op, EpiResXK<T>{x_half, x_proj, x_dual, tau, (T)d->rho}, kx, n, vop, gx, nullptr, ws, st);
      hipLaunchKernelGGL(admm_residual_fold_kernel, dim3(1), dim3(kBlock), 0, st, d->out4, static_cast<const double*>(d->workspace), gz, gx);
      PH_LAUNCH_END("admm residual fold");
    }
    default: set_error("admm_fused_stage: unknown stage"); return 1;
  }
}


// ---- pixel-ordered CG rounds: TWO launches per round (round 5) ------------------------------------------------------------
// For operators K = [D ; grad2d(nx, ny, L)] -- D couples the L channels of ONE pixel (row i of D has its L entries at columns
// i + c nx ny: the warp matrix [diag(Ix) diag(Iy)] of the TV-L1 flow shape, BASELINE config 4), D optional -- a thread that owns
// 4 consecutive pixels of an image column (2 in fp64) owns every row and every column of K that belongs to them: the D row, the
// 2 L gradient rows, the L primal entries.  What a stage needs from NEIGHBOURING pixels (the forward differences of the updated
// p at the right / lower neighbour, the backward differences of the updated r at the left / upper neighbour) it recomputes
// from the neighbour's operands instead of waiting for another thread to publish it -- the loads are cache hits -- so the
// vector updates of a round fold into the operator stages:
//   launch A (PQ):   beta, stopping test from the partials of the previous round ; p = beta p + s ; q = sqrt(Sigma) K sqrt(Tau) p ;
//                    |p|^2, |q|^2                                   [STEP_P2 + OP_FWD<FwdQ>;  t = sqrt(Tau) p is never stored]
//   launch B (XRS):  alpha ; x += alpha p ; r -= alpha q ; s = sqrt(Tau) (-shift x / sqrt(Tau) + K^T sqrt(Sigma) r) ;
//                    |x|^2, |s|^2                                   [STEP_XR2 + OP_ADJ<AdjS>; t = sqrt(Sigma) r is never stored]
// p and r are written to a second buffer each (a neighbour may still read the old values): round j reads p from P[(j-1) % 2]
// and leaves it in P[j % 2], reads r from R[j % 2] and leaves it in R[(j+1) % 2].  Per element every value is formed by the
// expressions of the four-launch round above in the same order (csr_rows / op_fwd_rows / op_adj_cols / the epilogues / StepXR /
// StepP), the sums are order-independent (reduce.hpp): x, p, q, r, s and every CG scalar are bit-identical to the four-launch
// round and to the staged round.  Per pixel and round (L = 2): A reads p, s, tau (6), D's values (2), sigma (5), writes p, q (7);
// B reads r, q (10), D's sigma (1), x, p, tau (6), D's values (2), writes r, x, s (9); A: sigma on D's rows only (1 instead of 5):
// 44 values instead of ~80.  (Sigma on the gradient rows is ONE number: the caller checks it.)
// D_CSR (round 6, prost_hip_pixel_op.d_csr): D is ANY sparse block with one row per pixel (a warp matrix: row i gathers the channels at
// pixels displaced from i).  The owner of pixel i still owns D's row i (launch A) and the L columns of D^T at i (launch B); the operand
// of an entry belongs to ANOTHER pixel and is recomputed from that pixel's stored operands like the stencil neighbours are -- three
// gathered values per entry instead of one, t never stored (csr_rows_formed below).
template <class T> struct PixArgs {
  unsigned nx, ny;                     // image; ny % VEC == 0
  size_t npx;                          // nx ny
  size_t d_row, g_row;                 // first row of the D block / of the gradient block
  const T* w;                          // D's values, row-major: w[i L + c] (the CSR value array of the block); nullptr-free when HAS_D
  // D_CSR (round 6): D is ANY CSR block of nx ny rows over the L nx ny primal entries (a warp matrix that gathers at displaced pixels):
  // its CSR arrays for the rows, those of D^T for the columns
  const T* d_val; const int32_t* d_ptr; const int32_t* d_ind;
  const T* dt_val; const int32_t* dt_ptr; const int32_t* dt_ind;
  const T* sigma; const T* tau;          // sigma: read on D's rows only
  T sig_g;                               // Sigma on the gradient rows: ONE value (a gradient block's row sums are constant, block_gradient2d.cu:154-158)
  T* x; const T* p_in; T* p_out; T* s; T* q; const T* r_in; T* r_out;
  T negshift;
  unsigned tiles;                      // workgroups
};
struct PixGeom { size_t px0; unsigned x, y0; bool active; };
template <int VEC>
__device__ __forceinline__ PixGeom pix_geom(unsigned tiles, unsigned ny, size_t npx) {
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (round-robin dispatch); each XCD takes a contiguous range of tiles, so
  // the neighbouring image columns a tile re-reads were fetched by the same XCD's L2 a moment ago
  unsigned t = blockIdx.x;
  if ((tiles & 7u) == 0) t = (blockIdx.x & 7u) * (tiles >> 3) + (blockIdx.x >> 3);
  PixGeom g;
  g.px0 = ((size_t)t * kBlock + threadIdx.x) * VEC;
  g.active = g.px0 < npx;
  const size_t c = g.active ? g.px0 : 0;
  g.x = (unsigned)(c / ny); g.y0 = (unsigned)(c - (size_t)g.x * ny);
  return g;
}

// sum[j] = sum_k val[k] f(ind[k]) over the CSR rows row0 .. row0 + V - 1 of this lane, entries in order (csr_rows of fused_op.hpp with the
// operand FORMED per entry: the vector the four-launch round would have stored is recomputed from its operands at the gathered position).
// `whole` (wave-uniform: the wavefront's 64 V rows from wave0 on all exist): the lanes take the rows TRANSPOSED -- lane, lane + 64, ... --
// so that neighbouring lanes read neighbouring row starts, entries and (for a warp) neighbouring gathered operands, and the sums are
// shuffled back to the lanes that own the rows (csr_contrib's scheme).  Taking a lane's own V rows instead costs 4-8 cache lines per
// lane and load: the first version of these instances ran launch A in 107 us at 1024^2 against 48 us for the two launches it replaces.
template <class T, int V, class F>
__device__ __forceinline__ void csr_rows_formed(const T* __restrict__ val, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ind, size_t row0, size_t wave0,
                                                bool whole, F f, T (&sum)[V]) {
  const unsigned lane = threadIdx.x & (kWave - 1);
  size_t r[V];
#pragma unroll
  for (int j = 0; j < V; j++) r[j] = whole ? wave0 + (size_t)j * kWave + lane : row0 + j;
  int32_t b[V], e[V], len = 0;
#pragma unroll
  for (int j = 0; j < V; j++) { b[j] = ptr[r[j]]; e[j] = ptr[r[j] + 1]; }
  T acc[V];
#pragma unroll
  for (int j = 0; j < V; j++) { acc[j] = 0; len = e[j] - b[j] > len ? e[j] - b[j] : len; }
  for (int32_t st = 0; st < len; st++) {
#pragma unroll
    for (int j = 0; j < V; j++) {
      const int32_t k = b[j] + st;
      if (k < e[j]) acc[j] += val[k] * f((size_t)ind[k]);
    }
  }
  if (V > 1 && whole) wave_untranspose<T, V>(acc, sum, lane);
  else {
#pragma unroll
    for (int j = 0; j < V; j++) sum[j] = acc[j];
  }
}

template <class T, int L, bool HAS_D, bool D_FIRST, bool FIRST, bool D_CSR = false>
__global__ void __launch_bounds__(kBlock, (L <= 2 ? 4 : 2)) cg_pixel_pq_kernel(PixArgs<T> a, const CgState* prev, CgState* cur, double* ws, RoundScalars sc) {
  constexpr int V = VecOf<T>::N;
  if (FIRST ? cur->done != 0 : prev->done != 0) {
    if (!FIRST && blockIdx.x == 0 && threadIdx.x == 0) *cur = *prev;
    return;
  }
  const PixGeom g = pix_geom<V>(a.tiles, a.ny, a.npx);
  const unsigned nx = a.nx, ny = a.ny;
  const bool right = g.active && g.x + 1 < nx, below = g.active && g.y0 + V < ny;
  // operands requested before the fold: none of them depends on beta
  T pc[L][V], sv[L][V], tc[L][V], pr[L][V], sr[L][V], tr[L][V], pb[L], sb_[L], tb[L];
#pragma unroll
  for (int c = 0; c < L; c++) {
#pragma unroll
    for (int j = 0; j < V; j++) { pc[c][j] = 0; sv[c][j] = 0; tc[c][j] = 1; pr[c][j] = 0; sr[c][j] = 0; tr[c][j] = 1; }
    pb[c] = 0; sb_[c] = 0; tb[c] = 1;
    const size_t e = (size_t)c * a.npx + g.px0;
    if (g.active) { ldv<T, V>(a.p_in + e, pc[c]); ldv<T, V>(a.tau + e, tc[c]); if (!FIRST) ldv<T, V>(a.s + e, sv[c]); }
    if (right) { ldv<T, V>(a.p_in + e + ny, pr[c]); ldv<T, V>(a.tau + e + ny, tr[c]); if (!FIRST) ldv<T, V>(a.s + e + ny, sr[c]); }
    if (below) { pb[c] = a.p_in[e + V]; tb[c] = a.tau[e + V]; if (!FIRST) sb_[c] = a.s[e + V]; }
  }
  T wv[HAS_D && !D_CSR ? V * L : 1], sgd[V];
  if (g.active && HAS_D) {
    if constexpr (!D_CSR) {
#pragma unroll
      for (int k = 0; k < L; k++) ldv<T, V>(a.w + g.px0 * L + (size_t)k * V, *reinterpret_cast<T(*)[V]>(&wv[k * V]));
    }
    ldv<T, V>(a.sigma + a.d_row + g.px0, sgd);
  }
  const T sqg = t_sqrt(a.sig_g);                       // sqrt(Sigma) of every gradient row
  T beta = 0;
  if (!FIRST) {
    // STEP_P2's head: beta and the stopping test from |s|^2, |x|^2 of the previous round (cgls.hpp:326-360); workgroup 0 records
    double s0, s1;
    fold_dd2(region(ws, kRegionS), sc.g_a, 2, region(ws, kRegionX), sc.g_b, 2, s0, s1);
    const double norms = sqrt(s0), gamma = norms * norms, normx = sqrt(s1);
    beta = (T)(gamma / prev->gamma);
    const bool done = (norms <= prev->norms0 * prev->tol) || (normx * prev->tol >= 1.);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const int indefinite = cur->indefinite;          // written by the previous round's launch B
      CgState r = *prev;
      r.indefinite = indefinite;
      r.norms = norms; r.gamma = gamma; r.beta = (double)beta; r.normx = normx;
      r.xmax = prev->xmax > normx ? prev->xmax : normx;
      if (done) {
        r.done = 1;
        if (sc.host_done) __hip_atomic_store(sc.host_done, prev->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        r.k = prev->k + 1;
      }
      *cur = r;
    }
    if (done) return;                                  // (the four-launch round still updates p here; nobody reads it afterwards)
  }
  dd_t sq{0.0, 0.0}, sp{0.0, 0.0};
  if (g.active) {
    // STEP_P: p = beta p + s ; t = sqrt(Tau) p -- own pixels (stored), right and lower neighbours (recomputed, not stored)
    T t0[L][V], t_r[L][V], t_b[L];
#pragma unroll
    for (int c = 0; c < L; c++) {
#pragma unroll
      for (int j = 0; j < V; j++) {
        if (!FIRST) { pc[c][j] = beta * pc[c][j] + sv[c][j]; pr[c][j] = beta * pr[c][j] + sr[c][j]; }
        t0[c][j] = t_sqrt(tc[c][j]) * pc[c][j];
        t_r[c][j] = t_sqrt(tr[c][j]) * pr[c][j];
        dd_acc(sp, (double)pc[c][j] * (double)pc[c][j]);
      }
      if (!FIRST) pb[c] = beta * pb[c] + sb_[c];
      t_b[c] = t_sqrt(tb[c]) * pb[c];
      if (!FIRST) stv<T, V>(a.p_out + (size_t)c * a.npx + g.px0, pc[c]);
    }
    // OP_FWD<FwdQ>: the D row (csr_rows: entries in order), then the gradient rows (op_fwd_rows), each scaled by EpiFwdQ
    if (HAS_D) {
      T qd[V], dsum[V];
      if constexpr (D_CSR) {
        // t = sqrt(Tau) p at the gathered primal entry, p updated as above (STEP_P's expressions; the stored p of that entry is written by its owner)
        const size_t wave0 = ((size_t)(g.px0 / V) - (threadIdx.x & (kWave - 1))) * V;          // first pixel of the wavefront
        csr_rows_formed<T, V>(a.d_val, a.d_ptr, a.d_ind, g.px0, wave0, wave0 + (size_t)kWave * V <= a.npx, [&](size_t col) {
          T pv = a.p_in[col];
          if (!FIRST) pv = beta * pv + a.s[col];
          return t_sqrt(a.tau[col]) * pv;
        }, dsum);
      }
#pragma unroll
      for (int j = 0; j < V; j++) {
        T sum = 0;
        if constexpr (D_CSR) sum = dsum[j];
        else {
#pragma unroll
          for (int c = 0; c < L; c++) sum += wv[j * L + c] * t0[c][j];
        }
        T kv = 0;
        kv = kv + sum;
        qd[j] = (T)1 * t_sqrt(sgd[j]) * kv;
        dd_acc(sq, (double)qd[j] * (double)qd[j]);
      }
      stv<T, V>(a.q + a.d_row + g.px0, qd);
    }
#pragma unroll
    for (int c = 0; c < L; c++) {
      T qx[V], qy[V];
#pragma unroll
      for (int j = 0; j < V; j++) {
        const T gx = g.x < nx - 1 ? t_r[c][j] - t0[c][j] : (T)0;
        const T dn = j + 1 < V ? t0[c][(j + 1) % V] : t_b[c];
        const T gy = g.y0 + j < ny - 1 ? dn - t0[c][j] : (T)0;
        T kx = 0, ky = 0;
        kx = kx + gx; ky = ky + gy;
        qx[j] = (T)1 * sqg * kx;
        qy[j] = (T)1 * sqg * ky;
        dd_acc(sq, (double)qx[j] * (double)qx[j]);
        dd_acc(sq, (double)qy[j] * (double)qy[j]);
      }
      stv<T, V>(a.q + a.g_row + (size_t)c * a.npx + g.px0, qx);
      stv<T, V>(a.q + a.g_row + (size_t)(L + c) * a.npx + g.px0, qy);
    }
  }
  block_dd_store1(sq, region(ws, kRegionQ), blockIdx.x);
  block_dd_store1(sp, region(ws, kRegionP), blockIdx.x);
}

template <class T, int L, bool HAS_D, bool D_FIRST, bool D_CSR = false>
__global__ void __launch_bounds__(kBlock, (L <= 2 ? 4 : 2)) cg_pixel_xrs_kernel(PixArgs<T> a, const CgState* cur, CgState* nxt, double* ws, RoundScalars sc) {
  constexpr int V = VecOf<T>::N;
  if (cur->done) return;
  const PixGeom g = pix_geom<V>(a.tiles, a.ny, a.npx);
  const unsigned nx = a.nx, ny = a.ny;
  const bool left = g.active && g.x > 0, above = g.active && g.y0 > 0;
  // operands requested before the fold (none depends on alpha): own rows of r, q, sigma; the d/dx rows of the left neighbour
  // column; the d/dy row of the pixel above; x, p, tau, D's values
  T rd[V], qd[V], gd[V];
  T rx[L][V], qx[L][V], ry[L][V], qy[L][V], rl[L][V], ql[L][V], ra[L], qa[L];
  T xv[L][V], pv[L][V], tv[L][V];
  T wv[HAS_D && !D_CSR ? V * L : 1];
#pragma unroll
  for (int j = 0; j < V; j++) { rd[j] = 0; qd[j] = 0; gd[j] = 1; }
#pragma unroll
  for (int c = 0; c < L; c++) {
#pragma unroll
    for (int j = 0; j < V; j++) { rl[c][j] = 0; ql[c][j] = 0; }
    ra[c] = 0; qa[c] = 0;
  }
  if (g.active) {
    if (HAS_D) {
      const size_t e = a.d_row + g.px0;
      ldv<T, V>(a.r_in + e, rd); ldv<T, V>(a.q + e, qd); ldv<T, V>(a.sigma + e, gd);
      if constexpr (!D_CSR) {
#pragma unroll
        for (int k = 0; k < L; k++) ldv<T, V>(a.w + g.px0 * L + (size_t)k * V, *reinterpret_cast<T(*)[V]>(&wv[k * V]));
      }
    }
#pragma unroll
    for (int c = 0; c < L; c++) {
      const size_t ex = a.g_row + (size_t)c * a.npx + g.px0, ey = a.g_row + (size_t)(L + c) * a.npx + g.px0;
      ldv<T, V>(a.r_in + ex, rx[c]); ldv<T, V>(a.q + ex, qx[c]);
      ldv<T, V>(a.r_in + ey, ry[c]); ldv<T, V>(a.q + ey, qy[c]);
      if (left) { ldv<T, V>(a.r_in + ex - ny, rl[c]); ldv<T, V>(a.q + ex - ny, ql[c]); }
      if (above) { ra[c] = a.r_in[ey - 1]; qa[c] = a.q[ey - 1]; }
    }
  }
  // STEP_XR2's head: alpha from |q|^2, |p|^2 (cgls.hpp:297-310)
  double s0, s1;
  fold_dd2(region(ws, kRegionQ), sc.g_a, 2, region(ws, kRegionP), sc.g_b, 2, s0, s1);
  const double normq = sqrt(s0), normp = sqrt(s1);
  double dlt = normq * normq + sc.shift * normp * normp;
  const int indefinite = dlt <= 0. ? 1 : 0;
  if (dlt == 0.) dlt = sc.eps;
  const T alpha = (T)(cur->gamma / dlt), neg_alpha = (T)(-cur->gamma / dlt);
  if (blockIdx.x == 0 && threadIdx.x == 0) nxt->indefinite = cur->indefinite | indefinite;
  dd_t sx{0.0, 0.0}, ss{0.0, 0.0};
  const T sqg = t_sqrt(a.sig_g);                       // sqrt(Sigma) of every gradient row
  if (g.active) {
    // second batch of operands (behind the fold's barriers: they arrive while r and t are formed; all at once would not fit 128 registers)
#pragma unroll
    for (int c = 0; c < L; c++) {
      const size_t en = (size_t)c * a.npx + g.px0;
      ldv<T, V>(a.x + en, xv[c]); ldv<T, V>(a.p_in + en, pv[c]); ldv<T, V>(a.tau + en, tv[c]);
    }
    // STEP_XR (m): r = -alpha q + r ; t = sqrt(Sigma) r -- own rows (stored), neighbour rows (recomputed, not stored)
    T td[V];
    if (HAS_D) {
#pragma unroll
      for (int j = 0; j < V; j++) { rd[j] = neg_alpha * qd[j] + rd[j]; td[j] = t_sqrt(gd[j]) * rd[j]; }
      stv<T, V>(a.r_out + a.d_row + g.px0, rd);
    }
#pragma unroll
    for (int c = 0; c < L; c++) {
      T tx[V], ty[V], tl[V];
#pragma unroll
      for (int j = 0; j < V; j++) {
        rx[c][j] = neg_alpha * qx[c][j] + rx[c][j]; tx[j] = sqg * rx[c][j];
        ry[c][j] = neg_alpha * qy[c][j] + ry[c][j]; ty[j] = sqg * ry[c][j];
        rl[c][j] = neg_alpha * ql[c][j] + rl[c][j]; tl[j] = sqg * rl[c][j];
      }
      ra[c] = neg_alpha * qa[c] + ra[c];
      const T t_above = sqg * ra[c];
      stv<T, V>(a.r_out + a.g_row + (size_t)c * a.npx + g.px0, rx[c]);
      stv<T, V>(a.r_out + a.g_row + (size_t)(L + c) * a.npx + g.px0, ry[c]);
      // STEP_XR (n): x = alpha p + x ; OP_ADJ<AdjS>: v = s0 ; + D^T t ; - div t (blocks in operator order) ; s = 1 sqrt(Tau) v
      T so[V], dcol[V];
      if constexpr (HAS_D && D_CSR) {
        // t = sqrt(Sigma) r at the gathered D row, r updated as above (STEP_XR's expressions; the stored r of that row is written by its owner)
        const size_t wave0 = ((size_t)(g.px0 / V) - (threadIdx.x & (kWave - 1))) * V;          // first pixel of the wavefront
        csr_rows_formed<T, V>(a.dt_val, a.dt_ptr, a.dt_ind, (size_t)c * a.npx + g.px0, (size_t)c * a.npx + wave0, wave0 + (size_t)kWave * V <= a.npx, [&](size_t row) {
          const size_t e = a.d_row + row;
          return t_sqrt(a.sigma[e]) * (neg_alpha * a.q[e] + a.r_in[e]);
        }, dcol);
      }
#pragma unroll
      for (int j = 0; j < V; j++) {
        xv[c][j] = alpha * pv[c][j] + xv[c][j];
        dd_acc(sx, (double)xv[c][j] * (double)xv[c][j]);
        const T sq = t_sqrt(tv[c][j]);
        T v = (a.negshift / ((T)1 * sq)) * xv[c][j];
        T dsum = 0;
        if constexpr (HAS_D && D_CSR) dsum = dcol[j];
        else if (HAS_D) dsum += wv[j * L + c] * td[j];
        T divy = g.y0 + j < ny - 1 ? ty[j] : (T)0;
        if (g.y0 + j > 0) divy -= j > 0 ? ty[(j + V - 1) % V] : t_above;
        T divx = g.x < nx - 1 ? tx[j] : (T)0;
        if (g.x > 0) divx -= tl[j];
        const T sdiv = divx + divy;
        if (HAS_D && D_FIRST) { v = v + dsum; v = v - sdiv; }
        else if (HAS_D) { v = v - sdiv; v = v + dsum; }
        else v = v - sdiv;
        so[j] = (T)1 * sq * v;
        dd_acc(ss, (double)so[j] * (double)so[j]);
      }
      stv<T, V>(a.x + (size_t)c * a.npx + g.px0, xv[c]);
      stv<T, V>(a.s + (size_t)c * a.npx + g.px0, so);
    }
  }
  block_dd_store1(sx, region(ws, kRegionX), blockIdx.x);
  block_dd_store1(ss, region(ws, kRegionS), blockIdx.x);
}

// the closing evaluation of a solve whose last queued round was round `last`: beta / stopping test of that round -> record last + 1
// (what launch A of round last + 1 would record), so that the result record (iterations, flags, norms) is that of the other paths
template <class T>
__global__ void __launch_bounds__(kBlock) cg_pixel_close_kernel(const CgState* prev, CgState* cur, double* ws, RoundScalars sc) {
  if (prev->done) { if (threadIdx.x == 0) *cur = *prev; return; }
  double s0, s1;
  fold_dd2(region(ws, kRegionS), sc.g_a, 2, region(ws, kRegionX), sc.g_b, 2, s0, s1);
  if (threadIdx.x != 0) return;
  const double norms = sqrt(s0), gamma = norms * norms, normx = sqrt(s1);
  CgState r = *prev;
  r.indefinite = cur->indefinite;
  r.norms = norms; r.gamma = gamma; r.beta = (double)(T)(gamma / prev->gamma); r.normx = normx;
  r.xmax = prev->xmax > normx ? prev->xmax : normx;
  if ((norms <= prev->norms0 * prev->tol) || (normx * prev->tol >= 1.)) {
    r.done = 1;
    if (sc.host_done) __hip_atomic_store(sc.host_done, prev->epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    r.k = prev->k + 1;
  }
  *cur = r;
}

static bool pixel_op_ok(const prost_hip_pixel_op* op, uint64_t m, uint64_t n, unsigned V) {
  if (!op || op->nx == 0 || op->ny == 0 || op->L < 1 || op->L > 3) return false;
  const uint64_t npx = op->nx * op->ny;
  if (op->ny % V || npx >= ((uint64_t)1 << 31)) return false;
  // one workgroup per tile of kBlock x V pixels writes a partial into the reduction workspace (cgls_pixel_round): images beyond
  // kReduceBlocks tiles (8.4 M pixels in fp32, 4.2 M in fp64 -- 4096^2) are refused HERE, so that BackendADMM::DescribeOperator falls
  // back to the four-launch rounds (whose fold grids are capped) instead of picking a path whose every solve then fails
  if ((npx / V + kBlock - 1) / kBlock > (uint64_t)kReduceBlocks) return false;
  if (n != (uint64_t)op->L * npx) return false;
  if (op->d_csr && (!op->has_d || !op->d_val || !op->d_ptr || !op->d_ind || !op->dt_val || !op->dt_ptr || !op->dt_ind)) return false;
  if (op->has_d) {
    if ((!op->d_csr && !op->w) || m != npx + 2 * (uint64_t)op->L * npx) return false;
    const bool d_first = op->d_row == 0 && op->g_row == npx, g_first = op->g_row == 0 && op->d_row == 2 * (uint64_t)op->L * npx;
    if (!d_first && !g_first) return false;
  } else if (m != 2 * (uint64_t)op->L * npx || op->g_row != 0) {
    return false;
  }
  return true;
}

template <class T>
static int cgls_pixel_round(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int round, int close, void* stream, void* const* ev4) {
  constexpr int V = VecOf<T>::N;
  if (!d || !d->state || !d->workspace) { set_error("cgls_pixel_round: state and workspace are required"); return 1; }
  if (round < 0) { set_error("cgls_pixel_round: negative round"); return 1; }
  if (!pixel_op_ok(op, d->m, d->n, V) || !op->p_alt || !op->r_alt) { set_error("cgls_pixel_round: unsupported operator description (prost_hip_pixel_op_supported)"); return 1; }
  T* P[2] = {static_cast<T*>(d->p), static_cast<T*>(op->p_alt)};
  T* R[2] = {static_cast<T*>(d->r), static_cast<T*>(op->r_alt)};
  for (const void* ptr : {(const void*)d->x, (const void*)d->q, (const void*)d->s, (const void*)d->sigma, (const void*)d->tau, op->d_csr ? (const void*)d->x : (const void*)op->w,
      (const void*)P[0], (const void*)P[1], (const void*)R[0], (const void*)R[1]})
    if (!aligned16(ptr)) { set_error("cgls_pixel_round: operands must be 16-byte aligned"); return 1; }
  hipStream_t st = as_stream(stream);
  const size_t npx = (size_t)(op->nx * op->ny);
  const unsigned tiles = (unsigned)((npx / V + kBlock - 1) / kBlock);
  if (tiles > (unsigned)kReduceBlocks) { set_error("cgls_pixel_round: image too large for the reduction workspace"); return 1; }
  // the partial sums of round 0's |p|^2 and of the solve's first |s|^2 come from the kernels of prost_hip_cgls_init_fused, whose grids
  // are what cgls_round computes for them; afterwards every region is written by `tiles` workgroups
  CgState* rec = static_cast<CgState*>(d->state);
  double* ws = static_cast<double*>(d->workspace);
  const double eps = (double)std::numeric_limits<T>::epsilon();
  PixArgs<T> a;
  a.nx = (unsigned)op->nx; a.ny = (unsigned)op->ny; a.npx = npx; a.d_row = (size_t)op->d_row; a.g_row = (size_t)op->g_row;
  a.w = static_cast<const T*>(op->w); a.sigma = static_cast<const T*>(d->sigma); a.tau = static_cast<const T*>(d->tau);
  a.d_val = static_cast<const T*>(op->d_val); a.d_ptr = op->d_ptr; a.d_ind = op->d_ind;
  a.dt_val = static_cast<const T*>(op->dt_val); a.dt_ptr = op->dt_ptr; a.dt_ind = op->dt_ind;
  const bool d_csr = op->has_d != 0 && op->d_csr != 0;
  a.x = static_cast<T*>(d->x); a.s = static_cast<T*>(d->s); a.q = static_cast<T*>(d->q);
  a.negshift = (T)(-d->shift); a.tiles = tiles; a.sig_g = (T)op->sigma_grad;
  const bool has_d = op->has_d != 0, d_first = has_d && op->d_first != 0;
  const int L = op->L;
  auto mark = [&](int k) { if (ev4 && ev4[2 * k] && ev4[2 * k + 1]) { g_launch_ev_start = (hipEvent_t)ev4[2 * k]; g_launch_ev_stop = (hipEvent_t)ev4[2 * k + 1]; } };
  if (close) {
    const RoundScalars sc{d->shift, eps, tiles, tiles, d->host_done};
    if (round < 1) { set_error("cgls_pixel_close: no round to close"); return 1; }
    PH_LAUNCH((cg_pixel_close_kernel<T>), dim3(1), dim3(kBlock), 0, st, rec + round - 1, rec + round, ws, sc);
    PH_LAUNCH_END("cgls pixel close");
  }
  // launch A of round j: beta / stopping test of round j - 1 -> record j ; p ; q
  a.p_in = round == 0 ? P[0] : P[(round - 1) & 1]; a.p_out = P[round & 1];
  a.r_in = R[round & 1]; a.r_out = R[(round + 1) & 1];
  const RoundScalars sa{d->shift, eps, tiles, tiles, d->host_done};
  mark(0);
#define PIX_A(LL, HD, DF, FI, DC) PH_LAUNCH((cg_pixel_pq_kernel<T, LL, HD, DF, FI, DC>), dim3(tiles), dim3(kBlock), 0, st, a, round == 0 ? rec : rec + round - 1, rec + round, ws, sa)
#define PIX_A_D(LL, FI) do { if (!has_d) PIX_A(LL, false, false, FI, false); else if (d_csr) { if (d_first) PIX_A(LL, true, true, FI, true); else PIX_A(LL, true, false, FI, true); } \
                             else if (d_first) PIX_A(LL, true, true, FI, false); else PIX_A(LL, true, false, FI, false); } while (0)
#define PIX_A_L(LL) do { if (round == 0) PIX_A_D(LL, true); else PIX_A_D(LL, false); } while (0)
  if (L == 1) PIX_A_L(1); else if (L == 2) PIX_A_L(2); else PIX_A_L(3);
#undef PIX_A_L
#undef PIX_A_D
#undef PIX_A
  // launch B of round j: alpha ; x, r ; s
  a.p_in = P[round & 1];
  const RoundScalars sb{d->shift, eps, tiles, tiles, nullptr};
  mark(1);
#define PIX_B(LL, HD, DF, DC) PH_LAUNCH((cg_pixel_xrs_kernel<T, LL, HD, DF, DC>), dim3(tiles), dim3(kBlock), 0, st, a, rec + round, rec + round + 1, ws, sb)
#define PIX_B_L(LL) do { if (!has_d) PIX_B(LL, false, false, false); else if (d_csr) { if (d_first) PIX_B(LL, true, true, true); else PIX_B(LL, true, false, true); } \
                         else if (d_first) PIX_B(LL, true, true, false); else PIX_B(LL, true, false, false); } while (0)
  if (L == 1) PIX_B_L(1); else if (L == 2) PIX_B_L(2); else PIX_B_L(3);
#undef PIX_B_L
#undef PIX_B
  PH_LAUNCH_END("cgls pixel round");
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {

size_t prost_hip_cgls_state_bytes(void) { return sizeof(CgState); }
size_t prost_hip_cgls_workspace_bytes(void) { return (size_t)kRegions * kReduceBlocks * 4 * sizeof(double); }
int prost_hip_cgls_stage_f32(int stage, const prost_hip_cgls_desc* d, void* stream) { return cgls_stage<float>(stage, d, stream); }
int prost_hip_cgls_stage_f64(int stage, const prost_hip_cgls_desc* d, void* stream) { return cgls_stage<double>(stage, d, stream); }

int prost_hip_admm_stage_f32(int stage, const prost_hip_admm_desc* d, void* stream) { return admm_stage<float>(stage, d, stream); }
int prost_hip_admm_stage_f64(int stage, const prost_hip_admm_desc* d, void* stream) { return admm_stage<double>(stage, d, stream); }

int prost_hip_normest_stage_f32(int stage, const prost_hip_normest_desc* d, void* stream) { return normest_stage<float>(stage, d, stream); }
int prost_hip_normest_stage_f64(int stage, const prost_hip_normest_desc* d, void* stream) { return normest_stage<double>(stage, d, stream); }

int prost_hip_fused_op_supported(const prost_hip_fused_op* op, uint64_t m, uint64_t n) { return fused_op_ok(op, m, n) ? 1 : 0; }
int prost_hip_cgls_round_f32(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, int round, void* stream) { return cgls_round<float>(d, op, round, stream); }
int prost_hip_cgls_round_f64(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, int round, void* stream) { return cgls_round<double>(d, op, round, stream); }
int prost_hip_cgls_round_timed_f32(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, int round, void* const* ev8, void* stream) { return cgls_round<float>(d, op, round, stream, ev8); }
int prost_hip_cgls_round_timed_f64(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, int round, void* const* ev8, void* stream) { return cgls_round<double>(d, op, round, stream, ev8); }
int prost_hip_cgls_init_fused_f32(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, void* stream) { return cgls_init_fused<float>(d, op, stream); }
int prost_hip_cgls_init_fused_f64(const prost_hip_cgls_desc* d, const prost_hip_fused_op* op, void* stream) { return cgls_init_fused<double>(d, op, stream); }
int prost_hip_admm_fused_stage_f32(int stage, const prost_hip_admm_desc* d, const prost_hip_fused_op* op, void* stream) { return admm_fused_stage<float>(stage, d, op, stream); }
int prost_hip_admm_fused_stage_f64(int stage, const prost_hip_admm_desc* d, const prost_hip_fused_op* op, void* stream) { return admm_fused_stage<double>(stage, d, op, stream); }

int prost_hip_pixel_op_supported(const prost_hip_pixel_op* op, uint64_t m, uint64_t n, int dtype) { return pixel_op_ok(op, m, n, dtype == 0 ? 4u : 2u) ? 1 : 0; }
int prost_hip_cgls_pixel_round_f32(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int round, void* stream) { return cgls_pixel_round<float>(d, op, round, 0, stream, nullptr); }
int prost_hip_cgls_pixel_round_f64(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int round, void* stream) { return cgls_pixel_round<double>(d, op, round, 0, stream, nullptr); }
int prost_hip_cgls_pixel_round_timed_f32(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int round, void* const* ev4,
    void* stream) { return cgls_pixel_round<float>(d, op, round, 0, stream, ev4); }
int prost_hip_cgls_pixel_round_timed_f64(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int round, void* const* ev4,
    void* stream) { return cgls_pixel_round<double>(d, op, round, 0, stream, ev4); }
int prost_hip_cgls_pixel_close_f32(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int last_round,
    void* stream) { return cgls_pixel_round<float>(d, op, last_round + 1, 1, stream, nullptr); }
int prost_hip_cgls_pixel_close_f64(const prost_hip_cgls_desc* d, const prost_hip_pixel_op* op, int last_round,
    void* stream) { return cgls_pixel_round<double>(d, op, last_round + 1, 1, stream, nullptr); }

int prost_hip_cgls_result(const void* state, prost_hip_cgls_result_t* out, void* stream) { return prost_hip_cgls_result_at(state, 0, out, stream); }
int prost_hip_cgls_result_at(const void* state, int index, prost_hip_cgls_result_t* out, void* stream) {
  CgState h;
  if (index < 0) { set_error("prost_hip_cgls_result_at: negative index"); return 1; }
  state = static_cast<const CgState*>(state) + index;
  PH_CHECK(hipMemcpyAsync(&h, state, sizeof(CgState), hipMemcpyDeviceToHost, as_stream(stream)));
  PH_CHECK(hipStreamSynchronize(as_stream(stream)));
  out->iterations = h.k; out->converged = h.done; out->indefinite = h.indefinite; out->flag = h.flag;
  out->norms = h.norms; out->norms0 = h.norms0; out->normx = h.normx; out->xmax = h.xmax;
  return 0;
}

}  // extern "C"
