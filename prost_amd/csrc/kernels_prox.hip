// kernels_prox.hip -- generic proximal-operator kernels for gfx950.
//
// One separable element per lane (prox_elem_operation.inl:59-94).  For the planar layout
// (interleaved = false) component i of element tx lives at tx + count*i: lanes of a wave read
// 64 consecutive values per component -> fully coalesced.  The function id is wave-uniform, so
// the 14-way dispatch is one scalar branch per wave (no per-function template explosion: the
// reference instantiates 176 kernels, prox_elem_operation.cu:44-256; this file compiles 4).
#include "common.hpp"
#include "device_math.hpp"

namespace prost_hip {

template <class T>
struct Coeffs {
  const T* ptr[7];
  T val[7];
};

template <class T, int OP>
__global__ void __launch_bounds__(kBlock) prox_elem_kernel(T* __restrict__ res, const T* __restrict__ arg,
                                                           const T* __restrict__ tau_diag, T tau_scal, bool invert_tau,
                                                           size_t count, size_t dim, bool interleaved, int fn, Coeffs<T> cf) {
  for (size_t tx = (size_t)blockIdx.x * kBlock + threadIdx.x; tx < count; tx += (size_t)gridDim.x * kBlock) {
    T c[7];
#pragma unroll
    for (int i = 0; i < 7; i++) c[i] = cf.ptr[i] ? cf.ptr[i][tx] : cf.val[i];
    if (OP == PROST_OP_1D) {
      // Vector index with dim = 1: both layouts give tx (vector.hpp:44-48)
      const T tau = elem_tau<T>(tau_scal, tau_diag[tx], invert_tau);
      res[tx] = elem_1d<T>(fn, arg[tx], tau, c);
    } else {
      // ElemOperationNorm2 (elem_operation_norm2.hpp:40-88)
      const size_t base = interleaved ? tx * dim : tx;
      const size_t stride = interleaved ? 1 : count;
      T norm = 0;
      for (size_t i = 0; i < dim; i++) { const T v = arg[base + i * stride]; norm += v * v; }
      if (norm > 0) {
        norm = t_sqrt(norm);
        const T tau = elem_tau<T>(tau_scal, tau_diag[base], invert_tau);   // tau_diag[0] only (:61)
        const T pr = scaled_prox<T>(fn, norm, tau, c);
        for (size_t i = 0; i < dim; i++) res[base + i * stride] = pr * arg[base + i * stride] / norm;
      } else {
        for (size_t i = 0; i < dim; i++) res[base + i * stride] = 0;
      }
    }
  }
}

template <class T>
static int launch_prox_elem(int op, int fn, T* res, const T* arg, const T* tau_diag, double tau, int invert, size_t count,
                            size_t dim, int interleaved, const T* const* coeff_ptr, const double* coeff_val, void* stream) {
  if (fn < 0 || fn >= PROST_FN_COUNT) { set_error("prox_elem: unknown function id"); return 1; }
  if (op != PROST_OP_1D && op != PROST_OP_NORM2) { set_error("prox_elem: unknown elem operation"); return 1; }
  if (count == 0) return 0;
  Coeffs<T> cf;
  for (int i = 0; i < 7; i++) { cf.ptr[i] = coeff_ptr ? coeff_ptr[i] : nullptr; cf.val[i] = (T)coeff_val[i]; }
  hipStream_t s = as_stream(stream);
  if (op == PROST_OP_1D)
    hipLaunchKernelGGL((prox_elem_kernel<T, PROST_OP_1D>), dim3(grid_for(count)), dim3(kBlock), 0, s, res, arg, tau_diag, (T)tau, invert != 0, count, (size_t)1, interleaved != 0, fn, cf);
  else
    hipLaunchKernelGGL((prox_elem_kernel<T, PROST_OP_NORM2>), dim3(grid_for(count)), dim3(kBlock), 0, s, res, arg, tau_diag, (T)tau, invert != 0, count, dim, interleaved != 0, fn, cf);
  PH_LAUNCH_END("prox_elem kernel");
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
template <class T>
__global__ void __launch_bounds__(kBlock) moreau_pre_kernel(T* __restrict__ out, const T* __restrict__ arg,
                                                            const T* __restrict__ td, T tau, bool inv, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    out[i] = inv ? arg[i] * (tau * td[i]) : arg[i] / (tau * td[i]);
}
template <class T>
__global__ void __launch_bounds__(kBlock) moreau_post_kernel(T* __restrict__ res, const T* __restrict__ arg,
                                                             const T* __restrict__ td, T tau, bool inv, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    res[i] = inv ? arg[i] - res[i] / (tau * td[i]) : arg[i] - tau * td[i] * res[i];
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_prox_elem_f32(int op, int fn, float* res, const float* arg, const float* td, double tau, int inv, size_t count, size_t dim, int il, const float* const* cp, const double* cv, void* s) {
  return launch_prox_elem<float>(op, fn, res, arg, td, tau, inv, count, dim, il, cp, cv, s);
}
int prost_hip_prox_elem_f64(int op, int fn, double* res, const double* arg, const double* td, double tau, int inv, size_t count, size_t dim, int il, const double* const* cp, const double* cv, void* s) {
  return launch_prox_elem<double>(op, fn, res, arg, td, tau, inv, count, dim, il, cp, cv, s);
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
  if (n == 0) return 0;
  hipLaunchKernelGGL((moreau_pre_kernel<float>), dim3(grid_for(n)), dim3(kBlock), 0, as_stream(s), o, a, td, (float)tau, inv != 0, n);
  PH_LAUNCH_END("moreau prescale");
}
int prost_hip_moreau_prescale_f64(double* o, const double* a, const double* td, double tau, int inv, size_t n, void* s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL((moreau_pre_kernel<double>), dim3(grid_for(n)), dim3(kBlock), 0, as_stream(s), o, a, td, tau, inv != 0, n);
  PH_LAUNCH_END("moreau prescale");
}
int prost_hip_moreau_postscale_f32(float* r, const float* a, const float* td, double tau, int inv, size_t n, void* s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL((moreau_post_kernel<float>), dim3(grid_for(n)), dim3(kBlock), 0, as_stream(s), r, a, td, (float)tau, inv != 0, n);
  PH_LAUNCH_END("moreau postscale");
}
int prost_hip_moreau_postscale_f64(double* r, const double* a, const double* td, double tau, int inv, size_t n, void* s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL((moreau_post_kernel<double>), dim3(grid_for(n)), dim3(kBlock), 0, as_stream(s), r, a, td, tau, inv != 0, n);
  PH_LAUNCH_END("moreau postscale");
}
}  // extern "C"
