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

#include "elementwise.hpp"

namespace prost_hip {

struct CgState {
  double gamma, norms0, norms, normx, xmax, alpha, neg_alpha, beta;
  double tol;                      // stopping tolerance and epoch of the current solve: set by INIT_X, so that the
  int epoch;                       // STEP launches take no per-solve argument (they can be replayed from a HIP graph)
  int done, k, indefinite, flag;
};

enum { kRegionX = 0, kRegionS, kRegionP, kRegionQ, kRegions };          // partial-sum regions of the workspace
__host__ __device__ inline double* region(double* ws, int r) { return ws + (size_t)r * 2 * kReduceBlocks; }

template <class T, int VEC, class F>
__global__ void __launch_bounds__(kBlock) cg_stage_kernel(F f, size_t n0, size_t n1, const CgState* st, double* ws) {
  if (F::kSkipWhenDone && st->done) return;
  f.load(st);
  double sa = 0, sb = 0;
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
    block_sum2_store(sa, sb, region(ws, F::kRegion), blockIdx.x);
    if (F::kRegion2 >= 0 && threadIdx.x == 0) region(ws, F::kRegion2)[2 * blockIdx.x] = region(ws, F::kRegion)[2 * blockIdx.x];
  }
}

// one workgroup: sum of the first component of `g` partial pairs, fixed association order
__device__ __forceinline__ double fold_region(const double* part, unsigned g) {
  __shared__ double s_w[kBlock / kWave];
  double a = 0;
  for (unsigned i = threadIdx.x; i < g; i += kBlock) a += part[2 * i];
  a = wave_sum(a);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = a;
  __syncthreads();
  double t = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; w++) t += s_w[w];
  return t;
}

enum { kScalarsInitX = 0, kScalarsInitS, kScalarsAlpha, kScalarsBeta };
struct ScalarArgs { double shift, tol, eps; unsigned g0, g1; int* host_done; int epoch; };

// the scalar recurrences of cgls.hpp, in double, values narrowed to T where the reference narrows them
template <class T, int WHICH>
__global__ void __launch_bounds__(kBlock) cg_scalar_kernel(CgState* stp, const double* ws, ScalarArgs a) {
  if (WHICH >= kScalarsAlpha && stp->done) return;
  const double s0 = fold_region(region(const_cast<double*>(ws), WHICH == kScalarsInitX ? kRegionX : WHICH == kScalarsAlpha ? kRegionQ : kRegionS), a.g0);
  const double s1 = WHICH == kScalarsAlpha ? fold_region(region(const_cast<double*>(ws), kRegionP), a.g1)
                  : WHICH == kScalarsBeta ? fold_region(region(const_cast<double*>(ws), kRegionX), a.g1) : 0.;
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
  template <int V> __device__ void range0(size_t i, double& sa, double&) const {
    T xv[V], dv[V], tv[V], sv[V];
    ldv<T, V>(x + i, xv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(dv[j]);
      tv[j] = sq * xv[j];
      sv[j] = (negshift / ((T)1 * sq)) * xv[j];
      sa += (double)xv[j] * (double)xv[j];
    }
    stv<T, V>(t + i, tv); stv<T, V>(s + i, sv);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// INIT_R (m):  r = (1 / (-1 sqrt(Sigma))) b   [gemv2 of r = b - A x on the copy r = b]
template <class T> struct InitR {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* b; const T* sigma; T* r;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, double&, double&) const {
    T bv[V], dv[V], rv[V];
    ldv<T, V>(b + i, bv); ldv<T, V>(sigma + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) rv[j] = ((T)1 / ((T)-1 * t_sqrt(dv[j]))) * bv[j];
    stv<T, V>(r + i, rv);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// INIT_R2 (m), after r += K t:  r = normx > 0 ? -1 sqrt(Sigma) r : b  (the reference skips the product for
//              x = 0, cgls.hpp:250);  t = sqrt(Sigma) r  [gemv1 of s = A'r - shift x]
template <class T> struct InitR2 {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* b; const T* sigma; T* r; T* t; bool nonzero;
  __device__ void load(const CgState* st) { nonzero = st->normx > 0.; }
  template <int V> __device__ void range0(size_t i, double&, double&) const {
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
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// INIT_S (n), after s += K^T t:  s = 1 sqrt(Tau) s;  p = s;  t = sqrt(Tau) p [gemv1 of q = A p];
//              partials of |s|^2 = |p|^2  (cgls.hpp:263-283)
template <class T> struct InitS {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = kRegionS, kRegion2 = kRegionP;
  T* s; T* p; T* t; const T* tau;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, double& sa, double&) const {
    T sv[V], dv[V], tv[V];
    ldv<T, V>(s + i, sv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(dv[j]);
      sv[j] = (T)1 * sq * sv[j];
      tv[j] = sq * sv[j];
      sa += (double)sv[j] * (double)sv[j];
    }
    stv<T, V>(s + i, sv); stv<T, V>(p + i, sv); stv<T, V>(t + i, tv);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// STEP_Q (m), after q = K t:  q = 1 sqrt(Sigma) q;  partials of |q|^2  (cgls.hpp:287-296)
template <class T> struct StepQ {
  static constexpr bool kSkipWhenDone = true, kTwoRanges = false;
  static constexpr int kRegion = kRegionQ, kRegion2 = -1;
  T* q; const T* sigma;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, double& sa, double&) const {
    T qv[V], dv[V];
    ldv<T, V>(q + i, qv); ldv<T, V>(sigma + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      qv[j] = (T)1 * t_sqrt(dv[j]) * qv[j];
      sa += (double)qv[j] * (double)qv[j];
    }
    stv<T, V>(q + i, qv);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// STEP_XR:  (n) x = alpha p + x;  s = (-shift / (1 sqrt(Tau))) x;  partials of |x|^2
//           (m) r = -alpha q + r;  t = sqrt(Sigma) r      (cgls.hpp:311-325, :352-354)
template <class T> struct StepXR {
  static constexpr bool kSkipWhenDone = true, kTwoRanges = true;
  static constexpr int kRegion = kRegionX, kRegion2 = -1;
  T* x; const T* p; T* s; const T* tau; T* r; const T* q; const T* sigma; T* t; T negshift;
  T alpha, neg_alpha;
  __device__ void load(const CgState* st) { alpha = (T)st->alpha; neg_alpha = (T)st->neg_alpha; }
  template <int V> __device__ void range0(size_t i, double& sa, double&) const {
    T xv[V], pv[V], dv[V], sv[V];
    ldv<T, V>(x + i, xv); ldv<T, V>(p + i, pv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      xv[j] = alpha * pv[j] + xv[j];
      sv[j] = (negshift / ((T)1 * t_sqrt(dv[j]))) * xv[j];
      sa += (double)xv[j] * (double)xv[j];
    }
    stv<T, V>(x + i, xv); stv<T, V>(s + i, sv);
  }
  template <int V> __device__ void range1(size_t i, double&, double&) const {
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
  template <int V> __device__ void range0(size_t i, double& sa, double&) const {
    T sv[V], dv[V];
    ldv<T, V>(s + i, sv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      sv[j] = (T)1 * t_sqrt(dv[j]) * sv[j];
      sa += (double)sv[j] * (double)sv[j];
    }
    stv<T, V>(s + i, sv);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// STEP_P (n):  p = beta p + s;  t = sqrt(Tau) p [gemv1 of the next q = A p];  partials of |p|^2  (cgls.hpp:341-351, :287-296)
template <class T> struct StepP {
  static constexpr bool kSkipWhenDone = true, kTwoRanges = false;
  static constexpr int kRegion = kRegionP, kRegion2 = -1;
  T* p; const T* s; T* t; const T* tau; T beta;
  __device__ void load(const CgState* st) { beta = (T)st->beta; }
  template <int V> __device__ void range0(size_t i, double& sa, double&) const {
    T pv[V], sv[V], dv[V], tv[V];
    ldv<T, V>(p + i, pv); ldv<T, V>(s + i, sv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      pv[j] = beta * pv[j] + sv[j];
      tv[j] = t_sqrt(dv[j]) * pv[j];
      sa += (double)pv[j] * (double)pv[j];
    }
    stv<T, V>(p + i, pv); stv<T, V>(t + i, tv);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
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
  template <int V> __device__ void range0(size_t i, double&, double&) const {
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
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// PRE_Z (m):  temp2 = sqrt(Sigma) (z_half + z_dual)                              temp2_functor :70-81
//             z_dual = (1 / (-1 sqrt(Sigma))) temp2                              gemv_functor2 of z_dual = temp2 - A temp1 (:399)
template <class T> struct AdmmPreZ {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* z_half; T* z_dual; const T* sigma; T* temp2;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, double&, double&) const {
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
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// PRE_Z2 (m), after z_dual += K temp3:  z_dual = -1 sqrt(Sigma) z_dual            gemv_functor3 of :399
template <class T> struct AdmmPreZ2 {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  T* z_dual; const T* sigma;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, double&, double&) const {
    T z[V], d[V];
    ldv<T, V>(z_dual + i, z); ldv<T, V>(sigma + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) z[j] = (T)-1 * t_sqrt(d[j]) * z[j];
    stv<T, V>(z_dual + i, z);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// POST_X (n), after the CG solve:  temp3 = x_proj (:437);  x_proj = sqrt(Tau) (x_proj + temp1)   x_proj_functor :447-456
template <class T> struct AdmmPostX {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = -1, kRegion2 = -1;
  T* x_proj; const T* temp1; const T* tau; T* temp3;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, double&, double&) const {
    T x[V], a[V], d[V], o[V];
    ldv<T, V>(x_proj + i, x); ldv<T, V>(temp1 + i, a); ldv<T, V>(tau + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) o[j] = t_sqrt(d[j]) * (x[j] + a[j]);
    stv<T, V>(temp3 + i, x); stv<T, V>(x_proj + i, o);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// POST_XZ, after z_proj = K x_proj:
//   (n) x_dual = temp1 sqrt(Tau) - x_proj  x_dual_functor :464-477;  temp1 = x_proj - x_dual  (prox_g argument :499)
//   (m) z_dual = temp2 / sqrt(Sigma) - z_proj  z_dual_functor :480-493;  temp2 = z_proj - z_dual  (prox_f argument :514)
template <class T> struct AdmmPostXZ {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = true;
  static constexpr int kRegion = -1, kRegion2 = -1;
  const T* x_proj; T* x_dual; T* temp1; const T* tau; const T* z_proj; T* z_dual; T* temp2; const T* sigma;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, double&, double&) const {
    T a[V], b[V], d[V], u[V], o[V];
    ldv<T, V>(temp1 + i, a); ldv<T, V>(x_proj + i, b); ldv<T, V>(tau + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      u[j] = a[j] * t_sqrt(d[j]) - b[j];
      o[j] = b[j] - u[j];
    }
    stv<T, V>(x_dual + i, u); stv<T, V>(temp1 + i, o);
  }
  template <int V> __device__ void range1(size_t i, double&, double&) const {
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
  template <int V> __device__ void range0(size_t i, double& sa, double& sb) const {
    T k[V], h[V], pj[V], du[V], d[V], y[V];
    ldv<T, V>(kx + i, k); ldv<T, V>(z_half + i, h); ldv<T, V>(z_proj + i, pj); ldv<T, V>(z_dual + i, du); ldv<T, V>(sigma + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      const T pr = sq * ((T)1.0 * k[j] + (T)-1.0 * h[j]);      // temp2 = -z_half; temp2 += K x_half; scaled
      const T pv = sq * h[j];
      sa += (double)pr * (double)pr;
      sb += (double)pv * (double)pv;
      y[j] = get_dual<T>(rho, d[j], (T)1, h[j], pj[j], du[j]);
    }
    stv<T, V>(kx + i, y);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// RES_X (n), with kty = K^T y:  w = get_dual(x_half, x_proj, x_dual, Tau, -1);  |sqrt(Tau) w| (dual variable norm, :570-590)
//             and |sqrt(Tau) (w + kty)| (dual residual, :596-616)
template <class T> struct AdmmResX {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = kRegionP, kRegion2 = -1;
  const T* kty; const T* x_half; const T* x_proj; const T* x_dual; const T* tau; T rho;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, double& sa, double& sb) const {
    T k[V], h[V], pj[V], du[V], d[V];
    ldv<T, V>(kty + i, k); ldv<T, V>(x_half + i, h); ldv<T, V>(x_proj + i, pj); ldv<T, V>(x_dual + i, du); ldv<T, V>(tau + i, d);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(d[j]);
      const T w = get_dual<T>(rho, d[j], (T)-1, h[j], pj[j], du[j]);
      const T dv = sq * w;
      const T dr = sq * ((T)1.0 * k[j] + w);
      sa += (double)dr * (double)dr;
      sb += (double)dv * (double)dv;
    }
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};

// out4 = {sqrt(sum RES_Z a), sqrt(sum RES_Z b), sqrt(sum RES_X a), sqrt(sum RES_X b)}
//      = {primal residual, primal variable norm, dual residual, dual variable norm}; out4 may be pinned host memory
__global__ void __launch_bounds__(kBlock) admm_residual_fold_kernel(double* out4, const double* ws, unsigned gz, unsigned gx) {
  const double* rz = region(const_cast<double*>(ws), kRegionQ);
  const double* rx = region(const_cast<double*>(ws), kRegionP);
  const double a = fold_region(rz, gz), b = fold_region(rz + 1, gz), c = fold_region(rx, gx), d = fold_region(rx + 1, gx);
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
  template <int V> __device__ void range0(size_t i, double&, double&) const {
    T xv[V], dv[V], o[V];
    ldv<T, V>(x + i, xv); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) { const T t = divide ? xv[j] / norm_x : xv[j]; o[j] = t_sqrt(dv[j]) * t; }
    stv<T, V>(x_temp + i, o);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// NORMEST_B (m), after ax = K x_temp:  a = sqrt(Sigma) ax;  partials of |a|^2;  ax = sqrt(Sigma) a
template <class T> struct NormestB {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = kRegionQ, kRegion2 = -1;
  T* ax; const T* sigma;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, double& sa, double&) const {
    T v[V], dv[V];
    ldv<T, V>(ax + i, v); ldv<T, V>(sigma + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) {
      const T sq = t_sqrt(dv[j]);
      const T a = sq * v[j];
      sa += (double)a * (double)a;
      v[j] = sq * a;
    }
    stv<T, V>(ax + i, v);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// NORMEST_C (n), after x_temp = K^T ax:  x = sqrt(Tau) x_temp;  partials of |x|^2
template <class T> struct NormestC {
  static constexpr bool kSkipWhenDone = false, kTwoRanges = false;
  static constexpr int kRegion = kRegionP, kRegion2 = -1;
  T* x; const T* x_temp; const T* tau;
  __device__ void load(const CgState*) {}
  template <int V> __device__ void range0(size_t i, double& sa, double&) const {
    T v[V], dv[V];
    ldv<T, V>(x_temp + i, v); ldv<T, V>(tau + i, dv);
#pragma unroll
    for (int j = 0; j < V; j++) { v[j] = t_sqrt(dv[j]) * v[j]; sa += (double)v[j] * (double)v[j]; }
    stv<T, V>(x + i, v);
  }
  template <int V> __device__ void range1(size_t, double&, double&) const {}
};
// out[0] = sqrt(sum of the first components of g partial pairs of region r); out may be pinned host memory
__global__ void __launch_bounds__(kBlock) sqrt_fold_kernel(double* out, const double* ws, int r, unsigned g) {
  const double a = fold_region(region(const_cast<double*>(ws), r), g);
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
static int launch_scalars(const prost_hip_cgls_desc* d, unsigned g0, unsigned g1, hipStream_t st) {
  ScalarArgs a{d->shift, d->tol, (double)std::numeric_limits<T>::epsilon(), g0, g1, d->host_done, d->epoch};
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

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {

size_t prost_hip_cgls_state_bytes(void) { return sizeof(CgState); }
size_t prost_hip_cgls_workspace_bytes(void) { return (size_t)kRegions * kReduceBlocks * 2 * sizeof(double); }
int prost_hip_cgls_stage_f32(int stage, const prost_hip_cgls_desc* d, void* stream) { return cgls_stage<float>(stage, d, stream); }
int prost_hip_cgls_stage_f64(int stage, const prost_hip_cgls_desc* d, void* stream) { return cgls_stage<double>(stage, d, stream); }

int prost_hip_admm_stage_f32(int stage, const prost_hip_admm_desc* d, void* stream) { return admm_stage<float>(stage, d, stream); }
int prost_hip_admm_stage_f64(int stage, const prost_hip_admm_desc* d, void* stream) { return admm_stage<double>(stage, d, stream); }

int prost_hip_normest_stage_f32(int stage, const prost_hip_normest_desc* d, void* stream) { return normest_stage<float>(stage, d, stream); }
int prost_hip_normest_stage_f64(int stage, const prost_hip_normest_desc* d, void* stream) { return normest_stage<double>(stage, d, stream); }

int prost_hip_cgls_result(const void* state, prost_hip_cgls_result_t* out, void* stream) {
  CgState h;
  PH_CHECK(hipMemcpyAsync(&h, state, sizeof(CgState), hipMemcpyDeviceToHost, as_stream(stream)));
  PH_CHECK(hipStreamSynchronize(as_stream(stream)));
  out->iterations = h.k; out->converged = h.done; out->indefinite = h.indefinite; out->flag = h.flag;
  out->norms = h.norms; out->norms0 = h.norms0; out->normx = h.normx; out->xmax = h.xmax;
  return 0;
}

}  // extern "C"
