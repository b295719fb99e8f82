// kernels_pdhg.hip -- generic (unfused) PDHG / ADMM / CGLS vector kernels for gfx950.
// Streaming elementwise work: grid-stride loops, consecutive lanes on consecutive addresses.
#include "elementwise.hpp"
#include "pdhg_rule.hpp"
#include "reduce.hpp"

namespace prost_hip {

__global__ void __launch_bounds__(kBlock) fold_partials_kernel(double* __restrict__ out, const double* __restrict__ partial,
                                                               unsigned nslots, bool sqrt_first) {
  double a = 0, b = 0;
  for (unsigned i = threadIdx.x; i < nslots; i += kBlock) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  __shared__ double s_a[kBlock / kWave], s_b[kBlock / kWave];
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) { s_a[wave] = a; s_b[wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0, tb = 0;
    for (int w = 0; w < kBlock / kWave; w++) { ta += s_a[w]; tb += s_b[w]; }
    out[0] = sqrt_first ? sqrt(ta) : ta;
    out[1] = tb;
  }
}

int launch_fold(double* out, const double* partial, unsigned nslots, bool sqrt_first, hipStream_t s) {
  hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(kBlock), 0, s, out, partial, nslots, sqrt_first);
  PH_LAUNCH_END("fold partials");
}

// stage 2 of TWO reductions in one launch: (out4[0], out4[1]) from partial1 and (out4[2], out4[3]) from partial2, each in the order
// of fold_partials_kernel -- and, on request, the step-size rule and the stopping test of the device record behind them
// (pdhg_rule.hpp): the five launches of a residual iteration on the generic path (two reductions, two folds, the rule) become two
template <class T>
__global__ void __launch_bounds__(kBlock) fold_pair_rule_kernel(double* __restrict__ out4, const double* __restrict__ partial1, unsigned n1,
                                                                const double* __restrict__ partial2, unsigned n2, PdhgRecord<T>* rec, int apply_rule,
                                                                unsigned long long iteration, prost_hip_pdhg_rule_state* mirror) {
  __shared__ double s_a[kBlock / kWave], s_b[kBlock / kWave];
  __shared__ double tot[4];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  for (int h = 0; h < 2; h++) {
    const double* partial = h == 0 ? partial1 : partial2;
    const unsigned nslots = h == 0 ? n1 : n2;
    double a = 0, b = 0;
    for (unsigned i = threadIdx.x; i < nslots; i += kBlock) { a += partial[2 * i]; b += partial[2 * i + 1]; }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) { s_a[wave] = a; s_b[wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double ta = 0, tb = 0;
      for (int w = 0; w < kBlock / kWave; w++) { ta += s_a[w]; tb += s_b[w]; }
      out4[2 * h] = ta; out4[2 * h + 1] = tb;
      tot[2 * h] = ta; tot[2 * h + 1] = tb;
    }
    __syncthreads();
  }
  if (apply_rule && rec && threadIdx.x == 0) rule_apply_device<T>(rec, tot, iteration, mirror);
}

// the residual sums of an iteration whose prox launches added them up themselves (kernels_prox.hip, operator sources): slots of 4 doubles
// (a.hi, a.lo, b.hi, b.lo; reduce.hpp) -- the primal sums from the launches of prox_f*, the dual sums from those of prox_g -- folded
// order-independently, and on request the step-size rule and the stopping test behind them (pdhg_rule.hpp)
// (1024 lanes: the first 8 wavefronts fold the primal slots, the other 8 the dual slots, every lane its slots in batches of four 32-byte
// loads -- one workgroup, two memory round trips for 2 x 4096 slots; with 256 lanes and the two regions one after the other this launch
// took 14.8 us per iteration at 2048^2)
constexpr int kFoldLanes = 1024;
template <class T>
__global__ void __launch_bounds__(kFoldLanes) fold_sums_rule_kernel(double* __restrict__ out4, const double* __restrict__ ws_primal, unsigned n_primal,
                                                                    const double* __restrict__ ws_dual, unsigned n_dual, PdhgRecord<T>* rec, int apply_rule,
                                                                    unsigned long long iteration, prost_hip_pdhg_rule_state* mirror) {
  if (rec && rec->stop) return;
  __shared__ dd_t s_part[kFoldLanes / kWave][2];
  __shared__ double tot[4];
  constexpr unsigned kHalf = kFoldLanes / 2;
  const unsigned side = threadIdx.x / kHalf, t = threadIdx.x % kHalf;
  const double* __restrict__ ws = side ? ws_dual : ws_primal;
  const unsigned n = side ? n_dual : n_primal;
  dd_t a{0.0, 0.0}, b{0.0, 0.0};
  typedef double d4 __attribute__((ext_vector_type(4)));
  for (unsigned base = t; base < n; base += 4 * kHalf) {
    d4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned i = base + k * kHalf;
      v[k] = i < n ? *reinterpret_cast<const d4*>(ws + 4 * (size_t)i) : d4{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { a = dd_add(a, dd_t{v[k][0], v[k][1]}); b = dd_add(b, dd_t{v[k][2], v[k][3]}); }
  }
  a = wave_sum_dd(a);
  b = wave_sum_dd(b);
  if ((threadIdx.x & (kWave - 1)) == 0) { s_part[threadIdx.x / kWave][0] = a; s_part[threadIdx.x / kWave][1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    constexpr int kW = kFoldLanes / kWave / 2;
    dd_t r[4] = {s_part[0][0], s_part[0][1], s_part[kW][0], s_part[kW][1]};
#pragma unroll
    for (int w = 1; w < kW; w++) {
      r[0] = dd_add(r[0], s_part[w][0]); r[1] = dd_add(r[1], s_part[w][1]);
      r[2] = dd_add(r[2], s_part[kW + w][0]); r[3] = dd_add(r[3], s_part[kW + w][1]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { out4[k] = r[k].hi; tot[k] = r[k].hi; }
    if (apply_rule && rec) rule_apply_device<T>(rec, tot, iteration, mirror);
  }
}
template <class T>
static int fold_sums(double* out4, const double* ws_primal, unsigned n_primal, const double* ws_dual, unsigned n_dual, void* record, int apply_rule,
                     unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!out4 || !ws_primal || !ws_dual) { set_error("pdhg_fold_sums: null argument"); return 1; }
  if ((reinterpret_cast<uintptr_t>(ws_primal) | reinterpret_cast<uintptr_t>(ws_dual)) & 31u) { set_error("pdhg_fold_sums: the slot arrays must start on 32-byte boundaries"); return 1; }
  hipLaunchKernelGGL(fold_sums_rule_kernel<T>, dim3(1), dim3(kFoldLanes), 0, as_stream(stream), out4, ws_primal, n_primal, ws_dual, n_dual, static_cast<PdhgRecord<T>*>(record),
                     apply_rule, iteration, mirror);
  PH_LAUNCH_END("fold sums kernel");
}

// ---- per-element formulas (one functor each; the skeletons in elementwise.hpp vectorise them) ----
// primal_proxarg_functor (backend_pdhg.cu:38-51): in = x, T, kty
// `rec`: the device-resident step-size record of a batch of iterations (prost_hip_use_step_record; fused_common.hpp: PdhgRecord) --
// the step sizes are then read from it on the device, the values passed by the host are ignored, and the map kernels return at once
// when its stop flag is raised
template <class T> struct PrimalArgF {
  T tau; const PdhgRecord<T>* rec;
  __device__ void prepare() { if (rec) tau = rec->p.tau; }
  __device__ bool skip() const { return rec && rec->stop; }
  __device__ T operator()(const T* a) const { return a[0] - tau * a[1] * a[2]; }
};
// dual_proxarg_functor (backend_pdhg.cu:54-70): in = y, S, kx, kx_prev
template <class T> struct DualArgF {
  T sigma, theta; const PdhgRecord<T>* rec;
  __device__ void prepare() { if (rec) { sigma = rec->p.sigma; theta = rec->p.theta; } }
  __device__ bool skip() const { return rec && rec->stop; }
  __device__ T operator()(const T* a) const { return a[0] + sigma * a[1] * ((1 + theta) * a[2] - theta * a[3]); }
};
// primal_residual_transform (backend_pdhg.cu:97-120): in = y_prev, y, S, kx_prev, kx
template <class T> struct ResidualPrimalF {
  T sigma, theta; const PdhgRecord<T>* rec;
  __device__ void prepare() { if (rec) { sigma = rec->p.sigma; theta = rec->p.theta; } }
  __device__ void operator()(const T* v, double& a, double& b) const {
    const T sd = v[2];
    const T z_hat = (v[0] - v[1]) / (sigma * t_sqrt(sd)) + t_sqrt(sd) * ((1 + theta) * v[4] - theta * v[3]);
    const T diff = z_hat - t_sqrt(sd) * v[4];
    a += (double)(diff * diff);
    b += (double)(z_hat * z_hat);
  }
};
// dual_residual_transform (backend_pdhg.cu:73-94): in = x_prev, x, T, kty_prev, kty
template <class T> struct ResidualDualF {
  T tau; const PdhgRecord<T>* rec;
  __device__ void prepare() { if (rec) tau = rec->p.tau; }
  __device__ void operator()(const T* v, double& a, double& b) const {
    const T td = v[2];
    const T w_hat = (v[0] - v[1]) / (tau * t_sqrt(td)) - t_sqrt(td) * v[3];
    const T diff = w_hat + t_sqrt(td) * v[4];
    a += (double)(diff * diff);
    b += (double)(w_hat * w_hat);
  }
};
// compute_w_variable_functor / compute_z_variable_functor (backend_pdhg.cu:147-186)
template <class T> struct WVarF { T tau; __device__ T operator()(const T* a) const { return (a[0] - a[1]) / (a[2] * tau) - a[3]; } };   // x_prev, x, T, kty_prev
template <class T> struct ZVarF {                                                                                                       // y_prev, y, S, kx, kx_prev
  T sigma, theta;
  __device__ T operator()(const T* a) const { return (a[0] - a[1]) / (sigma * a[2]) + (1 + theta) * a[3] - theta * a[4]; }
};

// comparison of two vectors (test / verification entry: prost_hip_compare_*): a = elements that differ in VALUE (what
// numpy.array_equal(..., equal_nan=True) counts: +0 == -0, NaN == NaN), b = sum |a - b| over them
template <class T> struct CompareF {
  __device__ void operator()(const T* v, double& a, double& b) const {
    const bool same = v[0] == v[1] || (v[0] != v[0] && v[1] != v[1]);
    if (!same) { a += 1.0; const double d = fabs((double)v[0] - (double)v[1]); b += d == d ? d : 0.0; }
  }
};

template <class T> struct FillF { T v; __device__ T operator()(const T*) const { return v; } };

// ---- ADMM / CGLS ----
template <class T> struct Nrm2F { __device__ void operator()(const T* v, double& a, double&) const { a += (double)v[0] * (double)v[0]; } };
template <class T> struct AxpyF { T alpha; __device__ T operator()(const T* a) const { return alpha * a[0] + a[1]; } };   // x, y
// backend_admm.cu:53-196; `a` holds the operands the op reads, in the order given at the launch below
template <class T, int OP> struct AdmmF {
  T alpha, beta;
  __device__ T operator()(const T* a) const {
    switch (OP) {
      case PROST_ADMM_TEMP1: return (alpha * a[0] + (1 - alpha) * a[1] + a[2]) / t_sqrt(a[3]);   // a b c d
      case PROST_ADMM_TEMP2: return t_sqrt(a[2]) * (a[0] + a[1]);                                // a b c
      case PROST_ADMM_DIFF: return a[0] - a[1];                                                  // a b
      case PROST_ADMM_XPROJ: return t_sqrt(a[2]) * (a[0] + a[1]);                                // o a b
      case PROST_ADMM_XDUAL: return a[0] * t_sqrt(a[2]) - a[1];                                  // a b c
      case PROST_ADMM_ZDUAL: return a[0] / t_sqrt(a[2]) - a[1];                                  // a b c
      case PROST_ADMM_GEMV1: return t_sqrt(a[0]) * a[1];                                         // a b
      case PROST_ADMM_GEMV2: return (beta / (alpha * t_sqrt(a[0]))) * a[1];                      // a b
      case PROST_ADMM_GEMV3: return alpha * t_sqrt(a[0]) * a[1];                                 // a b
      case PROST_ADMM_GETDUAL: {                                                                 // a b c d
        // get_dual_functor (backend_admm.cu:181-196) is called with the exponents +1 and -1 only: pow(x, 1) = x and pow(x, -1) = 1 / x
        // for a correctly rounded pow; the device's pow is not (a few results per thousand differ in the last place) -- kernels_cgls.hip, get_dual
        const T pw = beta == (T)1 ? a[3] : beta == (T)-1 ? (T)1 / a[3] : t_pow(a[3], beta);
        return -alpha * pw * (a[0] - a[1] + a[2]);
      }
      case PROST_ADMM_SCALE: return alpha * a[0];                                                // a
      default: return a[0] / alpha;                                                              // PROST_ADMM_DIV: a
    }
  }
};

template <class T>
static int launch_admm(int op, T* o, const T* a, const T* b, const T* c, const T* d, double alpha, double beta, size_t n, void* stream) {
  if (n == 0) return 0;
  hipStream_t s = as_stream(stream);
  const T al = (T)alpha, be = (T)beta;
#define GO(OPv, NIN, ...) case OPv: return launch_ew<T, NIN>("admm elem kernel", o, EwIn<T, NIN>{{__VA_ARGS__}}, n, AdmmF<T, OPv>{al, be}, s);
  switch (op) {
    GO(PROST_ADMM_TEMP1, 4, a, b, c, d) GO(PROST_ADMM_TEMP2, 3, a, b, c) GO(PROST_ADMM_DIFF, 2, a, b) GO(PROST_ADMM_XPROJ, 3, o, a, b)
    GO(PROST_ADMM_XDUAL, 3, a, b, c) GO(PROST_ADMM_ZDUAL, 3, a, b, c) GO(PROST_ADMM_GEMV1, 2, a, b) GO(PROST_ADMM_GEMV2, 2, a, b)
    GO(PROST_ADMM_GEMV3, 2, a, b) GO(PROST_ADMM_GETDUAL, 4, a, b, c, d) GO(PROST_ADMM_SCALE, 1, a) GO(PROST_ADMM_DIV, 1, a)
    default: set_error("admm_elem: unknown op"); return 1;
  }
#undef GO
}

template <class T, int NIN, class F>
static int reduce_to(double* out2, void* ws, const EwIn<T, NIN>& in, size_t n, F f, bool sqrt_first, void* stream) {
  hipStream_t st = as_stream(stream);
  if (n == 0) { PH_CHECK(hipMemsetAsync(out2, 0, (sqrt_first ? 1 : 2) * sizeof(double), st)); return 0; }
  double* partial = static_cast<double*>(ws);
  const unsigned g = launch_reduce2<T, NIN>(partial, in, n, f, st);
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "reduction kernel"); }
  return launch_fold(out2, partial, g, sqrt_first, st);
}

// |x|_2 with an order-independent sum (reduce.hpp, dd_t): the norms of cgls::Solve and of the ADMM residuals decide alpha, beta and
// the stopping tests -- every implementation of the solve (host-side scalars, device-resident stages, fused rounds) and the
// CPU oracle obtain the same double
template <class T, int VEC>
__global__ void __launch_bounds__(kBlock) nrm2_dd_kernel(const T* __restrict__ x, size_t n, double* __restrict__ partial) {
  dd_t a{0.0, 0.0};
  const size_t nv = n / VEC;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (size_t)gridDim.x * kBlock) {
    T v[VEC];
    ldv<T, VEC>(x + i * VEC, v);
#pragma unroll
    for (int j = 0; j < VEC; j++) dd_acc(a, (double)v[j] * (double)v[j]);
  }
  if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < n - nv * VEC) { const T v = x[nv * VEC + threadIdx.x]; dd_acc(a, (double)v * (double)v); }
  block_dd_store1(a, partial, blockIdx.x);
}
__global__ void __launch_bounds__(kBlock) nrm2_dd_fold_kernel(double* out, const double* partial, unsigned g) {
  const double t = fold_dd(partial, g, 2);
  if (threadIdx.x == 0) { out[0] = sqrt(t); out[1] = 0.0; }
}
template <class T>
static int nrm2_dd(double* out, const T* x, size_t n, void* ws, void* stream) {
  hipStream_t st = as_stream(stream);
  if (n == 0) { PH_CHECK(hipMemsetAsync(out, 0, 2 * sizeof(double), st)); return 0; }
  constexpr int V = VecOf<T>::N;
  const bool vec = aligned16(x) && n >= (size_t)V;
  unsigned g = grid_for(vec ? n / V : n, 2);
  if (g > (unsigned)kReduceBlocks) g = (unsigned)kReduceBlocks;
  double* partial = static_cast<double*>(ws);
  if (vec) hipLaunchKernelGGL((nrm2_dd_kernel<T, V>), dim3(g), dim3(kBlock), 0, st, x, n, partial);
  else hipLaunchKernelGGL((nrm2_dd_kernel<T, 1>), dim3(g), dim3(kBlock), 0, st, x, n, partial);
  hipLaunchKernelGGL(nrm2_dd_fold_kernel, dim3(1), dim3(kBlock), 0, st, out, partial, g);
  PH_LAUNCH_END("nrm2");
}

// bm = a == 0 ? sentinel : b ;  counts a outside {0, 1}   (prost_hip_mask_merge)
template <class T>
__global__ void __launch_bounds__(kBlock) mask_merge_kernel(T* __restrict__ bm, const T* __restrict__ a, const T* __restrict__ b, T b_val, size_t n,
                                                            unsigned long long* nonbinary) {
  unsigned long long bad = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const T av = a[i];
    if (av != (T)0 && av != (T)1) bad++;
    T out = b ? b[i] : b_val;
    if (av == (T)0) {
      if constexpr (sizeof(T) == 4) out = __uint_as_float(PROST_HIP_MASK_SENTINEL_F32);
      else out = __longlong_as_double((long long)PROST_HIP_MASK_SENTINEL_F64);
    }
    bm[i] = out;
  }
  if (bad) atomicAdd(nonbinary, bad);
}
template <class T>
static int launch_mask_merge(T* bm, const T* a, const T* b, double b_val, size_t n, unsigned long long* nonbinary, void* stream) {
  if (n == 0) return 0;
  if (!bm || !a || !nonbinary) { set_error("mask_merge: bm, a and the counter are required"); return 1; }
  hipLaunchKernelGGL((mask_merge_kernel<T>), dim3(grid_for(n, 4)), dim3(kBlock), 0, as_stream(stream), bm, a, b, (T)b_val, n, nonbinary);
  PH_LAUNCH_END("mask merge kernel");
}

template <class T>
__global__ void rule_after_sums_kernel(PdhgRecord<T>* rec, const double* sums4, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror) {
  rule_apply_device<T>(rec, sums4, iteration, mirror);
}

template <class T>
static int run_residuals(double* out4, const T* yp, const T* y, const T* S, const T* kxp, const T* kx, double sg, double th, size_t m, const T* xp, const T* x,
                         const T* Tr, const T* kp, const T* k, double tau, size_t n, void* ws, void* record, int apply_rule, unsigned long long iteration,
                         prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!out4 || !ws) { set_error("pdhg_residuals: output and workspace required"); return 1; }
  if (apply_rule && !record) { set_error("pdhg_residuals: the rule needs a record"); return 1; }
  constexpr int V = VecOf<T>::N;
  const EwIn<T, 5> in1{{yp, y, S, kxp, kx}}, in2{{xp, x, Tr, kp, k}};
  bool v1, v2;
  const unsigned g1 = reduce2_geometry<T, 5>(in1, m, v1), g2 = reduce2_geometry<T, 5>(in2, n, v2);
  if (m == 0 || n == 0 || (size_t)g1 + g2 > (size_t)kReduceBlocks) {
    // (a size without entries, or more workgroups than one workspace holds: the separate launches)
    if (int rc = reduce_to<T, 5>(out4, ws, in1, m, ResidualPrimalF<T>{(T)sg, (T)th, step_record<T>()}, false, stream)) return rc;
    if (int rc = reduce_to<T, 5>(out4 + 2, ws, in2, n, ResidualDualF<T>{(T)tau, step_record<T>()}, false, stream)) return rc;
    if (apply_rule) {
      hipLaunchKernelGGL(rule_after_sums_kernel<T>, dim3(1), dim3(1), 0, as_stream(stream), static_cast<PdhgRecord<T>*>(record), out4, iteration, mirror);
      PH_LAUNCH_END("pdhg rule kernel");
    }
    return 0;
  }
  hipStream_t st = as_stream(stream);
  double* p1 = static_cast<double*>(ws);
  double* p2 = p1 + 2 * (size_t)g1;
  const ResidualPrimalF<T> f1{(T)sg, (T)th, step_record<T>()};
  const ResidualDualF<T> f2{(T)tau, step_record<T>()};
#define GO(A, B) hipLaunchKernelGGL((reduce2_pair_kernel<T, A, B, 5, 5, ResidualPrimalF<T>, ResidualDualF<T>>), dim3(g1 + g2), dim3(kBlock), 0, st, p1, in1, m, f1, g1, p2, in2, n, f2)
  if (v1 && v2) GO(V, V); else if (v1) GO(V, 1); else if (v2) GO(1, V); else GO(1, 1);
#undef GO
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "residual reduction kernel"); }
  hipLaunchKernelGGL(fold_pair_rule_kernel<T>, dim3(1), dim3(kBlock), 0, st, out4, p1, g1, p2, g2, static_cast<PdhgRecord<T>*>(record), apply_rule, iteration, mirror);
  PH_LAUNCH_END("residual fold kernel");
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {

size_t prost_hip_reduce_workspace_bytes(void) { return (size_t)kReduceBlocks * 2 * sizeof(double); }

int prost_hip_pdhg_primal_arg_f32(float* t, const float* x, const float* T, const float* k, double tau, size_t n, void* s) {
  return launch_ew<float, 3>("primal_arg", t, EwIn<float, 3>{{x, T, k}}, n, PrimalArgF<float>{(float)tau, step_record<float>()}, as_stream(s));
}
int prost_hip_pdhg_primal_arg_f64(double* t, const double* x, const double* T, const double* k, double tau, size_t n, void* s) {
  return launch_ew<double, 3>("primal_arg", t, EwIn<double, 3>{{x, T, k}}, n, PrimalArgF<double>{tau, step_record<double>()}, as_stream(s));
}
int prost_hip_pdhg_dual_arg_f32(float* t, const float* y, const float* S, const float* kx, const float* kxp, double sg, double th, size_t m, void* s) {
  return launch_ew<float, 4>("dual_arg", t, EwIn<float, 4>{{y, S, kx, kxp}}, m, DualArgF<float>{(float)sg, (float)th, step_record<float>()}, as_stream(s));
}
int prost_hip_pdhg_dual_arg_f64(double* t, const double* y, const double* S, const double* kx, const double* kxp, double sg, double th, size_t m, void* s) {
  return launch_ew<double, 4>("dual_arg", t, EwIn<double, 4>{{y, S, kx, kxp}}, m, DualArgF<double>{sg, th, step_record<double>()}, as_stream(s));
}

int prost_hip_pdhg_residual_primal_f32(double* out2, const float* yp, const float* y, const float* S, const float* kxp, const float* kx, double sg, double th, size_t m, void* ws, void* s) {
  return reduce_to<float, 5>(out2, ws, EwIn<float, 5>{{yp, y, S, kxp, kx}}, m, ResidualPrimalF<float>{(float)sg, (float)th, step_record<float>()}, false, s);
}
int prost_hip_pdhg_residual_primal_f64(double* out2, const double* yp, const double* y, const double* S, const double* kxp, const double* kx, double sg, double th, size_t m, void* ws, void* s) {
  return reduce_to<double, 5>(out2, ws, EwIn<double, 5>{{yp, y, S, kxp, kx}}, m, ResidualPrimalF<double>{sg, th, step_record<double>()}, false, s);
}
int prost_hip_pdhg_residual_dual_f32(double* out2, const float* xp, const float* x, const float* T, const float* kp, const float* k, double tau, size_t n, void* ws, void* s) {
  return reduce_to<float, 5>(out2, ws, EwIn<float, 5>{{xp, x, T, kp, k}}, n, ResidualDualF<float>{(float)tau, step_record<float>()}, false, s);
}
int prost_hip_pdhg_residual_dual_f64(double* out2, const double* xp, const double* x, const double* T, const double* kp, const double* k, double tau, size_t n, void* ws, void* s) {
  return reduce_to<double, 5>(out2, ws, EwIn<double, 5>{{xp, x, T, kp, k}}, n, ResidualDualF<double>{tau, step_record<double>()}, false, s);
}

int prost_hip_pdhg_residuals_f32(double* out4, const float* yp, const float* y, const float* S, const float* kxp, const float* kx, double sg, double th, size_t m,
                                 const float* xp, const float* x, const float* T, const float* kp, const float* k, double tau, size_t n, void* ws, void* record,
                                 int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* s) {
  return run_residuals<float>(out4, yp, y, S, kxp, kx, sg, th, m, xp, x, T, kp, k, tau, n, ws, record, apply_rule, iteration, mirror, s);
}
int prost_hip_pdhg_residuals_f64(double* out4, const double* yp, const double* y, const double* S, const double* kxp, const double* kx, double sg, double th, size_t m,
                                 const double* xp, const double* x, const double* T, const double* kp, const double* k, double tau, size_t n, void* ws, void* record,
                                 int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* s) {
  return run_residuals<double>(out4, yp, y, S, kxp, kx, sg, th, m, xp, x, T, kp, k, tau, n, ws, record, apply_rule, iteration, mirror, s);
}

int prost_hip_pdhg_fold_sums_f32(double* out4, const double* ws_primal, unsigned n_primal, const double* ws_dual, unsigned n_dual, void* record, int apply_rule,
                                 unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* s) {
  return fold_sums<float>(out4, ws_primal, n_primal, ws_dual, n_dual, record, apply_rule, iteration, mirror, s);
}
int prost_hip_pdhg_fold_sums_f64(double* out4, const double* ws_primal, unsigned n_primal, const double* ws_dual, unsigned n_dual, void* record, int apply_rule,
                                 unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* s) {
  return fold_sums<double>(out4, ws_primal, n_primal, ws_dual, n_dual, record, apply_rule, iteration, mirror, s);
}
int prost_hip_pdhg_w_variable_f32(float* w, const float* xp, const float* x, const float* T, const float* kp, double tau, size_t n, void* s) {
  return launch_ew<float, 4>("w_variable", w, EwIn<float, 4>{{xp, x, T, kp}}, n, WVarF<float>{(float)tau}, as_stream(s));
}
int prost_hip_pdhg_w_variable_f64(double* w, const double* xp, const double* x, const double* T, const double* kp, double tau, size_t n, void* s) {
  return launch_ew<double, 4>("w_variable", w, EwIn<double, 4>{{xp, x, T, kp}}, n, WVarF<double>{tau}, as_stream(s));
}
int prost_hip_pdhg_z_variable_f32(float* z, const float* yp, const float* y, const float* S, const float* kx, const float* kxp, double sg, double th, size_t m, void* s) {
  return launch_ew<float, 5>("z_variable", z, EwIn<float, 5>{{yp, y, S, kx, kxp}}, m, ZVarF<float>{(float)sg, (float)th}, as_stream(s));
}
int prost_hip_pdhg_z_variable_f64(double* z, const double* yp, const double* y, const double* S, const double* kx, const double* kxp, double sg, double th, size_t m, void* s) {
  return launch_ew<double, 5>("z_variable", z, EwIn<double, 5>{{yp, y, S, kx, kxp}}, m, ZVarF<double>{sg, th}, as_stream(s));
}

int prost_hip_fill_f32(float* x, double value, size_t n, void* s) { return launch_ew<float, 0>("fill", x, EwIn<float, 0>{}, n, FillF<float>{(float)value}, as_stream(s)); }
int prost_hip_fill_f64(double* x, double value, size_t n, void* s) { return launch_ew<double, 0>("fill", x, EwIn<double, 0>{}, n, FillF<double>{value}, as_stream(s)); }
int prost_hip_compare_f32(double* out2, const float* a, const float* b, size_t n, void* ws, void* s) {
  return reduce_to<float, 2>(out2, ws, EwIn<float, 2>{{a, b}}, n, CompareF<float>{}, false, s);
}
int prost_hip_compare_f64(double* out2, const double* a, const double* b, size_t n, void* ws, void* s) {
  return reduce_to<double, 2>(out2, ws, EwIn<double, 2>{{a, b}}, n, CompareF<double>{}, false, s);
}

// fold writes out[0] = sqrt(sum), out[1] = 0 -> `out` must have room for 2 doubles
int prost_hip_nrm2_f32(double* out, const float* x, size_t n, void* ws, void* s) {
  return nrm2_dd<float>(out, x, n, ws, s);
}
int prost_hip_nrm2_f64(double* out, const double* x, size_t n, void* ws, void* s) {
  return nrm2_dd<double>(out, x, n, ws, s);
}
int prost_hip_axpy_f32(float* y, const float* x, double alpha, size_t n, void* s) {
  return launch_ew<float, 2>("axpy", y, EwIn<float, 2>{{x, y}}, n, AxpyF<float>{(float)alpha}, as_stream(s));
}
int prost_hip_axpy_f64(double* y, const double* x, double alpha, size_t n, void* s) {
  return launch_ew<double, 2>("axpy", y, EwIn<double, 2>{{x, y}}, n, AxpyF<double>{alpha}, as_stream(s));
}
int prost_hip_admm_elem_f32(int op, float* o, const float* a, const float* b, const float* c, const float* d, double alpha, double beta, size_t n, void* s) {
  return launch_admm<float>(op, o, a, b, c, d, alpha, beta, n, s);
}
int prost_hip_admm_elem_f64(int op, double* o, const double* a, const double* b, const double* c, const double* d, double alpha, double beta, size_t n, void* s) {
  return launch_admm<double>(op, o, a, b, c, d, alpha, beta, n, s);
}

int prost_hip_mask_merge_f32(float* bm, const float* a, const float* b, double b_val, size_t n, unsigned long long* nonbinary, void* s) { return launch_mask_merge<float>(bm, a, b, b_val, n, nonbinary, s); }
int prost_hip_mask_merge_f64(double* bm, const double* a, const double* b, double b_val, size_t n, unsigned long long* nonbinary, void* s) { return launch_mask_merge<double>(bm, a, b, b_val, n, nonbinary, s); }
}  // extern "C"
