// kernels_pdhg.hip -- generic (unfused) PDHG / ADMM / CGLS vector kernels for gfx950.
// Streaming elementwise work: grid-stride loops, consecutive lanes on consecutive addresses.
#include "common.hpp"
#include "device_math.hpp"
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

static inline unsigned reduce_grid(size_t n) {
  unsigned g = grid_for(n, 4);
  return g > (unsigned)kReduceBlocks ? (unsigned)kReduceBlocks : g;
}

// primal_proxarg_functor (backend_pdhg.cu:38-51)
template <class T>
__global__ void __launch_bounds__(kBlock) primal_arg_kernel(T* __restrict__ temp, const T* __restrict__ x, const T* __restrict__ Td,
                                                            const T* __restrict__ kty, T tau, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    temp[i] = x[i] - tau * Td[i] * kty[i];
}
// dual_proxarg_functor (backend_pdhg.cu:54-70)
template <class T>
__global__ void __launch_bounds__(kBlock) dual_arg_kernel(T* __restrict__ temp, const T* __restrict__ y, const T* __restrict__ Sd,
                                                          const T* __restrict__ kx, const T* __restrict__ kxp, T sigma, T theta, size_t m) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (size_t)gridDim.x * kBlock)
    temp[i] = y[i] + sigma * Sd[i] * ((1 + theta) * kx[i] - theta * kxp[i]);
}
// primal_residual_transform (backend_pdhg.cu:97-120)
template <class T>
__global__ void __launch_bounds__(kBlock) residual_primal_kernel(double* __restrict__ partial, const T* __restrict__ y_prev,
                                                                 const T* __restrict__ y, const T* __restrict__ Sd,
                                                                 const T* __restrict__ kxp, const T* __restrict__ kx,
                                                                 T sigma, T theta, size_t m) {
  double a = 0, b = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (size_t)gridDim.x * kBlock) {
    const T sd = Sd[i];
    const T z_hat = (y_prev[i] - y[i]) / (sigma * t_sqrt(sd)) + t_sqrt(sd) * ((1 + theta) * kx[i] - theta * kxp[i]);
    const T diff = z_hat - t_sqrt(sd) * kx[i];
    a += (double)(diff * diff);
    b += (double)(z_hat * z_hat);
  }
  block_sum2_store(a, b, partial, blockIdx.x);
}
// dual_residual_transform (backend_pdhg.cu:73-94)
template <class T>
__global__ void __launch_bounds__(kBlock) residual_dual_kernel(double* __restrict__ partial, const T* __restrict__ x_prev,
                                                               const T* __restrict__ x, const T* __restrict__ Td,
                                                               const T* __restrict__ ktyp, const T* __restrict__ kty, T tau, size_t n) {
  double a = 0, b = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const T td = Td[i];
    const T w_hat = (x_prev[i] - x[i]) / (tau * t_sqrt(td)) - t_sqrt(td) * ktyp[i];
    const T diff = w_hat + t_sqrt(td) * kty[i];
    a += (double)(diff * diff);
    b += (double)(w_hat * w_hat);
  }
  block_sum2_store(a, b, partial, blockIdx.x);
}
// compute_w_variable_functor / compute_z_variable_functor (backend_pdhg.cu:147-186)
template <class T>
__global__ void __launch_bounds__(kBlock) w_variable_kernel(T* __restrict__ w, const T* __restrict__ x_prev, const T* __restrict__ x,
                                                            const T* __restrict__ Td, const T* __restrict__ ktyp, T tau, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    w[i] = (x_prev[i] - x[i]) / (Td[i] * tau) - ktyp[i];
}
template <class T>
__global__ void __launch_bounds__(kBlock) z_variable_kernel(T* __restrict__ z, const T* __restrict__ y_prev, const T* __restrict__ y,
                                                            const T* __restrict__ Sd, const T* __restrict__ kx, const T* __restrict__ kxp,
                                                            T sigma, T theta, size_t m) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (size_t)gridDim.x * kBlock)
    z[i] = (y_prev[i] - y[i]) / (sigma * Sd[i]) + (1 + theta) * kx[i] - theta * kxp[i];
}

// ---- ADMM / CGLS ----
template <class T>
__global__ void __launch_bounds__(kBlock) nrm2_kernel(double* __restrict__ partial, const T* __restrict__ x, size_t n) {
  double a = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) a += (double)x[i] * (double)x[i];
  block_sum2_store(a, 0.0, partial, blockIdx.x);
}
template <class T>
__global__ void __launch_bounds__(kBlock) axpy_kernel(T* __restrict__ y, const T* __restrict__ x, T alpha, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) y[i] = alpha * x[i] + y[i];
}
template <class T, int OP>
__global__ void __launch_bounds__(kBlock) admm_elem_kernel(T* __restrict__ o, const T* __restrict__ a, const T* __restrict__ b,
                                                           const T* __restrict__ c, const T* __restrict__ d, T alpha, T beta, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    switch (OP) {   // backend_admm.cu:53-196
      case PROST_ADMM_TEMP1: o[i] = (alpha * a[i] + (1 - alpha) * b[i] + c[i]) / t_sqrt(d[i]); break;
      case PROST_ADMM_TEMP2: o[i] = t_sqrt(c[i]) * (a[i] + b[i]); break;
      case PROST_ADMM_DIFF: o[i] = a[i] - b[i]; break;
      case PROST_ADMM_XPROJ: o[i] = t_sqrt(b[i]) * (o[i] + a[i]); break;
      case PROST_ADMM_XDUAL: o[i] = a[i] * t_sqrt(c[i]) - b[i]; break;
      case PROST_ADMM_ZDUAL: o[i] = a[i] / t_sqrt(c[i]) - b[i]; break;
      case PROST_ADMM_GEMV1: o[i] = t_sqrt(a[i]) * b[i]; break;
      case PROST_ADMM_GEMV2: o[i] = (beta / (alpha * t_sqrt(a[i]))) * b[i]; break;
      case PROST_ADMM_GEMV3: o[i] = alpha * t_sqrt(a[i]) * b[i]; break;
      case PROST_ADMM_GETDUAL: o[i] = -alpha * t_pow(d[i], beta) * (a[i] - b[i] + c[i]); break;
      case PROST_ADMM_SCALE: o[i] = alpha * a[i]; break;
      case PROST_ADMM_DIV: o[i] = a[i] / alpha; break;
    }
  }
}

template <class T>
static int launch_admm(int op, T* o, const T* a, const T* b, const T* c, const T* d, double alpha, double beta, size_t n, void* stream) {
  if (n == 0) return 0;
  hipStream_t s = as_stream(stream);
  dim3 g(grid_for(n)), blk(kBlock);
#define GO(OPv) case OPv: hipLaunchKernelGGL((admm_elem_kernel<T, OPv>), g, blk, 0, s, o, a, b, c, d, (T)alpha, (T)beta, n); break;
  switch (op) {
    GO(PROST_ADMM_TEMP1) GO(PROST_ADMM_TEMP2) GO(PROST_ADMM_DIFF) GO(PROST_ADMM_XPROJ) GO(PROST_ADMM_XDUAL) GO(PROST_ADMM_ZDUAL)
    GO(PROST_ADMM_GEMV1) GO(PROST_ADMM_GEMV2) GO(PROST_ADMM_GEMV3) GO(PROST_ADMM_GETDUAL) GO(PROST_ADMM_SCALE) GO(PROST_ADMM_DIV)
    default: set_error("admm_elem: unknown op"); return 1;
  }
#undef GO
  PH_LAUNCH_END("admm elem kernel");
}

}  // namespace prost_hip

using namespace prost_hip;

#define ELEMWISE(n) if ((n) == 0) return 0; hipStream_t st = as_stream(s); dim3 g(grid_for(n)), blk(kBlock)

extern "C" {

size_t prost_hip_reduce_workspace_bytes(void) { return (size_t)kReduceBlocks * 2 * sizeof(double); }

int prost_hip_pdhg_primal_arg_f32(float* t, const float* x, const float* T, const float* k, double tau, size_t n, void* s) {
  ELEMWISE(n); hipLaunchKernelGGL((primal_arg_kernel<float>), g, blk, 0, st, t, x, T, k, (float)tau, n); PH_LAUNCH_END("primal_arg");
}
int prost_hip_pdhg_primal_arg_f64(double* t, const double* x, const double* T, const double* k, double tau, size_t n, void* s) {
  ELEMWISE(n); hipLaunchKernelGGL((primal_arg_kernel<double>), g, blk, 0, st, t, x, T, k, tau, n); PH_LAUNCH_END("primal_arg");
}
int prost_hip_pdhg_dual_arg_f32(float* t, const float* y, const float* S, const float* kx, const float* kxp, double sg, double th, size_t m, void* s) {
  ELEMWISE(m); hipLaunchKernelGGL((dual_arg_kernel<float>), g, blk, 0, st, t, y, S, kx, kxp, (float)sg, (float)th, m); PH_LAUNCH_END("dual_arg");
}
int prost_hip_pdhg_dual_arg_f64(double* t, const double* y, const double* S, const double* kx, const double* kxp, double sg, double th, size_t m, void* s) {
  ELEMWISE(m); hipLaunchKernelGGL((dual_arg_kernel<double>), g, blk, 0, st, t, y, S, kx, kxp, sg, th, m); PH_LAUNCH_END("dual_arg");
}

#define RESIDUAL_BODY(KERNEL, n, ...)                                                         \
  hipStream_t st = as_stream(s);                                                              \
  double* partial = static_cast<double*>(ws);                                                 \
  if ((n) == 0) { PH_CHECK(hipMemsetAsync(out2, 0, 2 * sizeof(double), st)); return 0; }      \
  const unsigned gsz = reduce_grid(n);                                                        \
  hipLaunchKernelGGL(KERNEL, dim3(gsz), dim3(kBlock), 0, st, partial, __VA_ARGS__);           \
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "residual"); }   \
  return launch_fold(out2, partial, gsz, false, st)

int prost_hip_pdhg_residual_primal_f32(double* out2, const float* yp, const float* y, const float* S, const float* kxp, const float* kx, double sg, double th, size_t m, void* ws, void* s) {
  RESIDUAL_BODY((residual_primal_kernel<float>), m, yp, y, S, kxp, kx, (float)sg, (float)th, m);
}
int prost_hip_pdhg_residual_primal_f64(double* out2, const double* yp, const double* y, const double* S, const double* kxp, const double* kx, double sg, double th, size_t m, void* ws, void* s) {
  RESIDUAL_BODY((residual_primal_kernel<double>), m, yp, y, S, kxp, kx, sg, th, m);
}
int prost_hip_pdhg_residual_dual_f32(double* out2, const float* xp, const float* x, const float* T, const float* kp, const float* k, double tau, size_t n, void* ws, void* s) {
  RESIDUAL_BODY((residual_dual_kernel<float>), n, xp, x, T, kp, k, (float)tau, n);
}
int prost_hip_pdhg_residual_dual_f64(double* out2, const double* xp, const double* x, const double* T, const double* kp, const double* k, double tau, size_t n, void* ws, void* s) {
  RESIDUAL_BODY((residual_dual_kernel<double>), n, xp, x, T, kp, k, tau, n);
}

int prost_hip_pdhg_w_variable_f32(float* w, const float* xp, const float* x, const float* T, const float* kp, double tau, size_t n, void* s) {
  ELEMWISE(n); hipLaunchKernelGGL((w_variable_kernel<float>), g, blk, 0, st, w, xp, x, T, kp, (float)tau, n); PH_LAUNCH_END("w_variable");
}
int prost_hip_pdhg_w_variable_f64(double* w, const double* xp, const double* x, const double* T, const double* kp, double tau, size_t n, void* s) {
  ELEMWISE(n); hipLaunchKernelGGL((w_variable_kernel<double>), g, blk, 0, st, w, xp, x, T, kp, tau, n); PH_LAUNCH_END("w_variable");
}
int prost_hip_pdhg_z_variable_f32(float* z, const float* yp, const float* y, const float* S, const float* kx, const float* kxp, double sg, double th, size_t m, void* s) {
  ELEMWISE(m); hipLaunchKernelGGL((z_variable_kernel<float>), g, blk, 0, st, z, yp, y, S, kx, kxp, (float)sg, (float)th, m); PH_LAUNCH_END("z_variable");
}
int prost_hip_pdhg_z_variable_f64(double* z, const double* yp, const double* y, const double* S, const double* kx, const double* kxp, double sg, double th, size_t m, void* s) {
  ELEMWISE(m); hipLaunchKernelGGL((z_variable_kernel<double>), g, blk, 0, st, z, yp, y, S, kx, kxp, sg, th, m); PH_LAUNCH_END("z_variable");
}

int prost_hip_nrm2_f32(double* out, const float* x, size_t n, void* ws, void* s) {
  hipStream_t st = as_stream(s); double* partial = static_cast<double*>(ws);
  if (n == 0) { PH_CHECK(hipMemsetAsync(out, 0, sizeof(double), st)); return 0; }
  const unsigned gsz = reduce_grid(n);
  hipLaunchKernelGGL((nrm2_kernel<float>), dim3(gsz), dim3(kBlock), 0, st, partial, x, n);
  // fold writes out[0] = sqrt(sum), out[1] = 0 -> `out` must have room for 2 doubles
  return launch_fold(out, partial, gsz, true, st);
}
int prost_hip_nrm2_f64(double* out, const double* x, size_t n, void* ws, void* s) {
  hipStream_t st = as_stream(s); double* partial = static_cast<double*>(ws);
  if (n == 0) { PH_CHECK(hipMemsetAsync(out, 0, sizeof(double), st)); return 0; }
  const unsigned gsz = reduce_grid(n);
  hipLaunchKernelGGL((nrm2_kernel<double>), dim3(gsz), dim3(kBlock), 0, st, partial, x, n);
  return launch_fold(out, partial, gsz, true, st);
}
int prost_hip_axpy_f32(float* y, const float* x, double alpha, size_t n, void* s) {
  ELEMWISE(n); hipLaunchKernelGGL((axpy_kernel<float>), g, blk, 0, st, y, x, (float)alpha, n); PH_LAUNCH_END("axpy");
}
int prost_hip_axpy_f64(double* y, const double* x, double alpha, size_t n, void* s) {
  ELEMWISE(n); hipLaunchKernelGGL((axpy_kernel<double>), g, blk, 0, st, y, x, alpha, n); PH_LAUNCH_END("axpy");
}
int prost_hip_admm_elem_f32(int op, float* o, const float* a, const float* b, const float* c, const float* d, double alpha, double beta, size_t n, void* s) {
  return launch_admm<float>(op, o, a, b, c, d, alpha, beta, n, s);
}
int prost_hip_admm_elem_f64(int op, double* o, const double* a, const double* b, const double* c, const double* d, double alpha, double beta, size_t n, void* s) {
  return launch_admm<double>(op, o, a, b, c, d, alpha, beta, n, s);
}

}  // extern "C"
