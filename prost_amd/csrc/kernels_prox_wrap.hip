// kernels_prox_wrap.hip -- the wrapper / projection proximal operators next to the elem-operation
// family: ProxTransform, ProxPermute, ProxIndHalfspace, ProxIndSOC, ProxIndSum and the
// elem-operation ind_sum.  All are single streaming passes: one lane per element (transform,
// permute) or per group with the planar layout (component i of group t at t + count*i), so the
// lanes of a wave read consecutive addresses per component.
#include "elementwise.hpp"

namespace prost_hip {

template <class T>
struct Coeff5 {
  const T* ptr[5];
  T val[5];
};

// ProxTransformPrescaleArgument + ProxTransformPrescaleStepSize (prox_transform.cu:27-78) in ONE pass:
// both read tau_diag and the coefficient vectors a, e; 16 bytes per lane when everything is aligned
template <class T, int VEC>
__global__ void __launch_bounds__(kBlock) transform_prescale_kernel(T* __restrict__ scaled_arg, T* __restrict__ scaled_tau,
                                                                    const T* __restrict__ arg, const T* __restrict__ tau_diag,
                                                                    Coeff5<T> cf, T tau, bool invert, size_t n) {
  const size_t nv = n / VEC;
  auto one = [&](T x, T td, T a, T b, T c, T d, T e, T& oa, T& ot) {
    T tau2 = tau * td;
    if (invert) tau2 = 1 / tau2;
    oa = (a * (x - tau2 * d)) / (1 + tau2 * e) - b;
    ot = (a * a * c * tau2) / (1 + tau2 * e);
  };
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (size_t)gridDim.x * kBlock) {
    T x[VEC], td[VEC], c[5][VEC], oa[VEC], ot[VEC];
    ldv<T, VEC>(arg + i * VEC, x);
    ldv<T, VEC>(tau_diag + i * VEC, td);
#pragma unroll
    for (int k = 0; k < 5; k++) {
      if (cf.ptr[k]) ldv<T, VEC>(cf.ptr[k] + i * VEC, c[k]);
      else {
#pragma unroll
        for (int j = 0; j < VEC; j++) c[k][j] = cf.val[k];
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) one(x[j], td[j], c[0][j], c[1][j], c[2][j], c[3][j], c[4][j], oa[j], ot[j]);
    stv<T, VEC>(scaled_arg + i * VEC, oa);
    stv<T, VEC>(scaled_tau + i * VEC, ot);
  }
  if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < n - nv * VEC) {
    const size_t i = nv * VEC + threadIdx.x;
    T c[5];
#pragma unroll
    for (int k = 0; k < 5; k++) c[k] = cf.ptr[k] ? cf.ptr[k][i] : cf.val[k];
    one(arg[i], tau_diag[i], c[0], c[1], c[2], c[3], c[4], scaled_arg[i], scaled_tau[i]);
  }
}

template <class T>
static int launch_transform_prescale(T* sa, T* st, const T* arg, const T* td, const T* const* cp, const double* cv, double tau, int invert,
                                     size_t n, void* stream) {
  if (n == 0) return 0;
  Coeff5<T> cf;
  bool vec = aligned16(sa) && aligned16(st) && aligned16(arg) && aligned16(td);
  for (int k = 0; k < 5; k++) { cf.ptr[k] = cp ? cp[k] : nullptr; cf.val[k] = (T)cv[k]; vec = vec && aligned16(cf.ptr[k]); }
  constexpr int V = VecOf<T>::N;
  if (vec && n >= (size_t)V) hipLaunchKernelGGL((transform_prescale_kernel<T, V>), dim3(grid_for(n / V)), dim3(kBlock), 0, as_stream(stream), sa, st, arg, td, cf, (T)tau, invert != 0, n);
  else hipLaunchKernelGGL((transform_prescale_kernel<T, 1>), dim3(grid_for(n)), dim3(kBlock), 0, as_stream(stream), sa, st, arg, td, cf, (T)tau, invert != 0, n);
  PH_LAUNCH_END("transform prescale kernel");
}

// ProxTransformPostscale (prox_transform.cu:80-97): in = result (in place), a, b
template <class T> struct TransformPostF { __device__ T operator()(const T* v) const { return (v[0] + v[2]) / v[1]; } };
template <class T> struct TransformPostAF { T b; __device__ T operator()(const T* v) const { return (v[0] + b) / v[1]; } };
template <class T> struct TransformPostBF { T a; __device__ T operator()(const T* v) const { return (v[0] + v[1]) / a; } };
template <class T> struct TransformPostSF { T a, b; __device__ T operator()(const T* v) const { return (v[0] + b) / a; } };

template <class T>
static int launch_transform_postscale(T* r, const T* ap, double av, const T* bp, double bv, size_t n, void* stream) {
  hipStream_t s = as_stream(stream);
  if (ap && bp) return launch_ew<T, 3>("transform postscale", r, EwIn<T, 3>{{r, ap, bp}}, n, TransformPostF<T>{}, s);
  if (ap) return launch_ew<T, 2>("transform postscale", r, EwIn<T, 2>{{r, ap}}, n, TransformPostAF<T>{(T)bv}, s);
  if (bp) return launch_ew<T, 2>("transform postscale", r, EwIn<T, 2>{{r, bp}}, n, TransformPostBF<T>{(T)av}, s);
  return launch_ew<T, 1>("transform postscale", r, EwIn<T, 1>{{r}}, n, TransformPostSF<T>{(T)av, (T)bv}, s);
}

// ProxPermuteKernel (prox_permute.cu:31-48)
template <class T>
__global__ void __launch_bounds__(kBlock) permute_kernel(T* __restrict__ res, const T* __restrict__ arg, const int32_t* __restrict__ perm,
                                                         size_t n, bool inverse) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    if (inverse) res[perm[i]] = arg[i];
    else res[i] = arg[perm[i]];
  }
}

// ProxIndHalfspaceKernel + ProjectHalfspace (prox_ind_halfspace.cu:31-86): always planar; a per group
// (count*dim, planar) or one normal of dim entries
template <class T>
__global__ void __launch_bounds__(kBlock) halfspace_kernel(T* __restrict__ res, const T* __restrict__ arg, size_t count, size_t dim,
                                                           const T* __restrict__ a, bool a_per_group, const T* __restrict__ b, bool b_per_group) {
  for (size_t tx = (size_t)blockIdx.x * kBlock + threadIdx.x; tx < count; tx += (size_t)gridDim.x * kBlock) {
    const T t = b_per_group ? b[tx] : b[0];
    T sq_norm = 0, iprod = 0;
    for (size_t i = 0; i < dim; i++) {
      const T n = a_per_group ? a[tx + count * i] : a[i];
      sq_norm += n * n;
      iprod += n * arg[tx + count * i];
    }
    const T excess = iprod - t;
    const T f = (excess > (T)0 ? excess : (T)0) / sq_norm;        // max(0, <n,v> - t) / |n|^2
    for (size_t i = 0; i < dim; i++) {
      const T n = a_per_group ? a[tx + count * i] : a[i];
      res[tx + count * i] = arg[tx + count * i] - f * n;
    }
  }
}

// ProxIndSOCKernel (prox_ind_soc.cu:30-77): (x_1 .. x_{dim-1}, y) planar, alpha == 1
template <class T>
__global__ void __launch_bounds__(kBlock) soc_kernel(T* __restrict__ res, const T* __restrict__ arg, size_t count, size_t dim) {
  for (size_t tx = (size_t)blockIdx.x * kBlock + threadIdx.x; tx < count; tx += (size_t)gridDim.x * kBlock) {
    const T y0 = arg[count * (dim - 1) + tx];
    T norm_x0 = 0;
    for (size_t i = 0; i + 1 < dim; i++) { const T v = arg[tx + count * i]; norm_x0 += v * v; }
    norm_x0 = t_sqrt(norm_x0);
    if (norm_x0 <= y0) {
      for (size_t i = 0; i + 1 < dim; i++) res[tx + count * i] = arg[tx + count * i];
      res[count * (dim - 1) + tx] = y0;
    } else if (norm_x0 <= -y0) {
      for (size_t i = 0; i + 1 < dim; i++) res[tx + count * i] = 0;
      res[count * (dim - 1) + tx] = 0;
    } else {
      const T fac = (y0 + norm_x0) / (2 * norm_x0);
      for (size_t i = 0; i + 1 < dim; i++) res[tx + count * i] = fac * arg[tx + count * i];
      res[count * (dim - 1) + tx] = fac * norm_x0;
    }
  }
}

// ProxIndSumKernel (prox_ind_sum.cu:30-66); res already holds a copy of arg (:119)
template <class T>
__global__ void __launch_bounds__(kBlock) ind_sum_kernel(T* __restrict__ res, const T* __restrict__ arg, const T* __restrict__ tau_diag,
                                                         const uint64_t* __restrict__ inds, size_t count, size_t dim, T total_sum, T tau, bool inv) {
  for (size_t tx = (size_t)blockIdx.x * kBlock + threadIdx.x; tx < count; tx += (size_t)gridDim.x * kBlock) {
    T sum_arg = 0, sum_tau = 0;
    for (size_t i = 0; i < dim; i++) {
      const size_t k = inds[tx * dim + i];
      T mytau = tau_diag[k] * tau;
      if (inv) mytau = (T)(1. / (double)mytau);
      sum_arg += arg[k];
      sum_tau += mytau;
    }
    for (size_t i = 0; i < dim; i++) {
      const size_t k = inds[tx * dim + i];
      T mytau = tau_diag[k] * tau;
      if (inv) mytau = (T)(1. / (double)mytau);
      res[k] = arg[k] - mytau * (sum_arg - total_sum) / sum_tau;
    }
  }
}

// ElemOperationIndSum (elem_operation_ind_sum.hpp:41-60)
template <class T>
__global__ void __launch_bounds__(kBlock) elem_ind_sum_kernel(T* __restrict__ res, const T* __restrict__ arg, size_t count, size_t dim,
                                                              bool interleaved) {
  for (size_t tx = (size_t)blockIdx.x * kBlock + threadIdx.x; tx < count; tx += (size_t)gridDim.x * kBlock) {
    const size_t base = interleaved ? tx * dim : tx, stride = interleaved ? 1 : count;
    T tl = 0;
    for (size_t i = 0; i < dim; i++) tl += arg[base + i * stride];
    tl = (T)(((double)tl - 1.) / (double)static_cast<T>(dim));
    for (size_t i = 0; i < dim; i++) res[base + i * stride] = arg[base + i * stride] - tl;
  }
}

// ElemOperationIndSimplex (elem_operation_ind_simplex.hpp:40-119).  The reference sorts in a 1024-entry
// per-thread local array; here the sort runs in a caller-provided workspace with the same PLANAR
// layout as the data (entry i of group tx at tx + count*i), so the lanes of a wave touch
// consecutive addresses whenever they are at the same sort position, and dim is not limited.
template <class T>
__global__ void __launch_bounds__(kBlock) elem_ind_simplex_kernel(T* __restrict__ res, const T* __restrict__ arg, T* __restrict__ work,
                                                                  size_t count, size_t dim, bool interleaved) {
  for (size_t tx = (size_t)blockIdx.x * kBlock + threadIdx.x; tx < count; tx += (size_t)gridDim.x * kBlock) {
    const size_t base = interleaved ? tx * dim : tx, stride = interleaved ? 1 : count;
    T* a = work + tx;                                          // a[i] == a[i * count]
    for (size_t i = 0; i < dim; i++) a[i * count] = arg[base + i * stride];
    const int gaps[6] = {132, 57, 23, 10, 4, 1};               // ShellSort :97-115, descending
    for (int k = 0; k < 6; k++) {
      const long gap = gaps[k];
      for (long i = gap; i < (long)dim; i++) {
        const T temp = a[i * count];
        long j = i;
        for (; (j >= gap) && (a[(j - gap) * count] <= temp); j -= gap) a[j * count] = a[(j - gap) * count];
        a[j * count] = temp;
      }
    }
    bool bget = false;
    T tmpsum = 0, tmax = 0;
    for (long ii = 1; ii <= (long)dim - 1; ii++) {
      tmpsum += a[(ii - 1) * count];
      tmax = (T)(((double)tmpsum - 1.) / (double)(T)ii);
      if (tmax >= a[ii * count]) { bget = true; break; }
    }
    if (!bget) tmax = (T)(((double)(T)(tmpsum + a[(dim - 1) * count]) - 1.0) / (double)(T)dim);
    for (size_t i = 0; i < dim; i++) {
      const T v = arg[base + i * stride] - tmax;
      res[base + i * stride] = v > (T)0 ? v : (T)0;            // max(val - tmax, 0): first argument wins on NaN
    }
  }
}

}  // namespace prost_hip

using namespace prost_hip;

#define GRID_LAUNCH(KERNEL, n, name, ...)                                                               \
  if ((n) == 0) return 0;                                                                               \
  hipLaunchKernelGGL(KERNEL, dim3(grid_for(n)), dim3(kBlock), 0, as_stream(s), __VA_ARGS__);            \
  PH_LAUNCH_END(name)

extern "C" {
int prost_hip_transform_prescale_f32(float* sa, float* st, const float* arg, const float* td, const float* const* cp, const double* cv, double tau, int inv, size_t n, void* s) {
  return launch_transform_prescale<float>(sa, st, arg, td, cp, cv, tau, inv, n, s);
}
int prost_hip_transform_prescale_f64(double* sa, double* st, const double* arg, const double* td, const double* const* cp, const double* cv, double tau, int inv, size_t n, void* s) {
  return launch_transform_prescale<double>(sa, st, arg, td, cp, cv, tau, inv, n, s);
}
int prost_hip_transform_postscale_f32(float* r, const float* ap, double av, const float* bp, double bv, size_t n, void* s) { return launch_transform_postscale<float>(r, ap, av, bp, bv, n, s); }
int prost_hip_transform_postscale_f64(double* r, const double* ap, double av, const double* bp, double bv, size_t n, void* s) { return launch_transform_postscale<double>(r, ap, av, bp, bv, n, s); }

int prost_hip_permute_f32(float* res, const float* arg, const int32_t* perm, size_t n, int inverse, void* s) { GRID_LAUNCH((permute_kernel<float>), n, "permute kernel", res, arg, perm, n, inverse != 0); }
int prost_hip_permute_f64(double* res, const double* arg, const int32_t* perm, size_t n, int inverse, void* s) { GRID_LAUNCH((permute_kernel<double>), n, "permute kernel", res, arg, perm, n, inverse != 0); }

int prost_hip_prox_ind_halfspace_f32(float* res, const float* arg, size_t count, size_t dim, const float* a, size_t sz_a, const float* b, size_t sz_b, void* s) {
  GRID_LAUNCH((halfspace_kernel<float>), count, "ind_halfspace kernel", res, arg, count, dim, a, sz_a == count * dim, b, sz_b == count);
}
int prost_hip_prox_ind_halfspace_f64(double* res, const double* arg, size_t count, size_t dim, const double* a, size_t sz_a, const double* b, size_t sz_b, void* s) {
  GRID_LAUNCH((halfspace_kernel<double>), count, "ind_halfspace kernel", res, arg, count, dim, a, sz_a == count * dim, b, sz_b == count);
}
int prost_hip_prox_ind_soc_f32(float* res, const float* arg, size_t count, size_t dim, void* s) { GRID_LAUNCH((soc_kernel<float>), count, "ind_soc kernel", res, arg, count, dim); }
int prost_hip_prox_ind_soc_f64(double* res, const double* arg, size_t count, size_t dim, void* s) { GRID_LAUNCH((soc_kernel<double>), count, "ind_soc kernel", res, arg, count, dim); }
int prost_hip_prox_ind_sum_f32(float* res, const float* arg, const float* td, const uint64_t* inds, size_t count, size_t dim, double total, double tau, int inv, void* s) {
  GRID_LAUNCH((ind_sum_kernel<float>), count, "ind_sum kernel", res, arg, td, inds, count, dim, (float)total, (float)tau, inv != 0);
}
int prost_hip_prox_ind_sum_f64(double* res, const double* arg, const double* td, const uint64_t* inds, size_t count, size_t dim, double total, double tau, int inv, void* s) {
  GRID_LAUNCH((ind_sum_kernel<double>), count, "ind_sum kernel", res, arg, td, inds, count, dim, total, tau, inv != 0);
}
int prost_hip_prox_elem_ind_sum_f32(float* res, const float* arg, size_t count, size_t dim, int il, void* s) { GRID_LAUNCH((elem_ind_sum_kernel<float>), count, "elem ind_sum kernel", res, arg, count, dim, il != 0); }
int prost_hip_prox_elem_ind_sum_f64(double* res, const double* arg, size_t count, size_t dim, int il, void* s) { GRID_LAUNCH((elem_ind_sum_kernel<double>), count, "elem ind_sum kernel", res, arg, count, dim, il != 0); }
int prost_hip_prox_elem_ind_simplex_f32(float* res, const float* arg, float* work, size_t count, size_t dim, int il, void* s) {
  if (dim == 0) return 0;
  GRID_LAUNCH((elem_ind_simplex_kernel<float>), count, "elem ind_simplex kernel", res, arg, work, count, dim, il != 0);
}
int prost_hip_prox_elem_ind_simplex_f64(double* res, const double* arg, double* work, size_t count, size_t dim, int il, void* s) {
  if (dim == 0) return 0;
  GRID_LAUNCH((elem_ind_simplex_kernel<double>), count, "elem ind_simplex kernel", res, arg, work, count, dim, il != 0);
}
}  // extern "C"
