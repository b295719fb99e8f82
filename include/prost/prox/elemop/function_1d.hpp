// prost/prox/elemop/function_1d.hpp -- the 14 scalar proximal maps behind sum_1d / sum_norm2, as functors.
//
// Plugin contract of the reference's include/prost/prox/elemop/function_1d.hpp:34-326: Function1D*<T> with
// `T operator()(T x0, T tau, T alpha, T beta) const` = argmin_x  tau f(x) + (x - x0)^2 / 2, composable with
// ElemOperation1D / ElemOperationNorm2 (elem_operation_1d.hpp, elem_operation_norm2.hpp) or a user's own operation.
// A user-written function is any struct with that call operator.
//
// This header IS the implementation the library's own kernels use (prost_amd/csrc/device_math.hpp includes it): the
// fp32 instantiations keep the reference's implicit promotions to double wherever a double literal appears in its
// expressions, so results round as the reference's do (tests: bit-exact against the compiled reference functors, `lq`
// within 5e-5 -- Newton on pow()).  Device header (hipcc).
#ifndef PROST_PROX_ELEMOP_FUNCTION_1D_HPP_
#define PROST_PROX_ELEMOP_FUNCTION_1D_HPP_
#include <hip/hip_runtime.h>

#include <cmath>

namespace prost {
namespace elemop {

template <class T> __host__ __device__ __forceinline__ T t_abs(T v) { return v < 0 ? -v : v; }
template <> __host__ __device__ __forceinline__ float t_abs<float>(float v) { return fabsf(v); }
template <> __host__ __device__ __forceinline__ double t_abs<double>(double v) { return fabs(v); }
__host__ __device__ __forceinline__ float t_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__host__ __device__ __forceinline__ double t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__host__ __device__ __forceinline__ float t_sqrt(float v) { return sqrtf(v); }
__host__ __device__ __forceinline__ double t_sqrt(double v) { return sqrt(v); }
__host__ __device__ __forceinline__ float t_pow(float a, float b) { return powf(a, b); }
__host__ __device__ __forceinline__ double t_pow(double a, double b) { return pow(a, b); }
__host__ __device__ __forceinline__ float t_sin(float v) { return sinf(v); }
__host__ __device__ __forceinline__ double t_sin(double v) { return sin(v); }
__host__ __device__ __forceinline__ float t_cos(float v) { return cosf(v); }
__host__ __device__ __forceinline__ double t_cos(double v) { return cos(v); }
__host__ __device__ __forceinline__ float t_acos(float v) { return acosf(v); }
__host__ __device__ __forceinline__ double t_acos(double v) { return acos(v); }

// x / d with the exact shortcut for d == 1
__host__ __device__ __forceinline__ double div1(double x, double d) { return d == 1.0 ? x : x / d; }
__host__ __device__ __forceinline__ float div1(float x, float d) { return d == 1.0f ? x : x / d; }

// ---- the scalar maps (function_1d.hpp, line ranges of the reference per function) ----
template <class T> __host__ __device__ __forceinline__ T f1d_abs(T x0, T tau) {            // :47-60
  if (x0 >= tau) return x0 - tau;
  if (x0 <= -tau) return x0 + tau;
  return (T)0;
}
template <class T> __host__ __device__ __forceinline__ T f1d_square(T x0, T tau) {         // :63-72
  return (T)div1((double)x0, 1. + (double)tau);
}
template <class T> __host__ __device__ __forceinline__ T f1d_l0(T x0, T tau) {             // :146-158
  return (x0 * x0 > 2 * tau) ? x0 : (T)0;
}
template <class T> __host__ __device__ inline T lq_newton(T t0, T alpha, T q, T eps) {     // :173-191
  T t = t0, delta = 0;
  int guard = 0;   // the reference loop has no bound; 200 Newton steps is far past convergence
  do {
    const T power = t_pow(t, q);
    const T dF1 = t - 1 + alpha * q * power / t;
    const T dF2 = 1 + alpha * q * (q - 1) * power / (t * t);
    delta = dF1 / dF2;
    t = t - delta;
  } while (delta > eps && ++guard < 200);
  return t;
}
template <class T> __host__ __device__ inline T lq_half(T alpha) {                          // :195-202
  const T sqrt3 = t_sqrt((T)3);
  const T PI_half = (T)1.5707963267948966192313216916397514420985846996875529;
  const T s = 2 * (t_sin((T)((t_acos((T)(alpha * 3 * sqrt3 / 4)) + PI_half) / 3))) / sqrt3;
  return s * s;
}
template <class T> __host__ __device__ __forceinline__ T lq_eps();
template <> __host__ __device__ __forceinline__ float lq_eps<float>() { return (float)1e-5; }   // :263-267
template <> __host__ __device__ __forceinline__ double lq_eps<double>() { return 1e-11; }        // :270-274

template <class T> __host__ __device__ inline T f1d_lq(T x0, T tau, T alpha, T beta) {      // :205-260
  if (alpha == 1) return f1d_abs(x0, tau);
  if (alpha == 0) return f1d_l0(x0, tau);
  T t = 0;
  if (t_abs(x0) > 0) {
    T factor = tau * t_pow(t_abs(x0), (T)(alpha - 2));
    if (alpha < 1) {
      const T t2 = 2 * (alpha - 1) / (alpha - 2);
      if ((double)factor < 0.5 * (double)(1 - (t2 - 1) * (t2 - 1)) / (double)t_pow(t2, alpha)) {
        if ((double)alpha == 0.5) t = lq_half<T>(factor);
        else t = lq_newton<T>((T)1, factor, alpha, lq_eps<T>());
      }
    } else {
      t = lq_newton<T>((T)1, factor, alpha, lq_eps<T>());
    }
  }
  return t * t_abs(x0);
}
template <class T> __host__ __device__ __forceinline__ T f1d_ind_leq0(T x0) { return x0 > (T)0 ? (T)0 : x0; }                   // :75-87
template <class T> __host__ __device__ __forceinline__ T f1d_ind_geq0(T x0) { return x0 < (T)0 ? (T)0 : x0; }                   // :90-102
template <class T> __host__ __device__ __forceinline__ T f1d_ind_box01(T x0) { return x0 > (T)1 ? (T)1 : (x0 < (T)0 ? (T)0 : x0); }   // :117-131
template <class T> __host__ __device__ __forceinline__ T f1d_max_pos0(T x0, T tau) { return x0 > tau ? x0 - tau : (x0 < (T)0 ? x0 : (T)0); }   // :134-148
template <class T> __host__ __device__ __forceinline__ T f1d_huber(T x0, T tau, T alpha) {                                       // :161-171
  T r = (T)(((double)(x0 / tau)) / (1. + (double)(alpha / tau)));
  const T ar = t_abs(r);
  r /= ((T)1 > ar ? (T)1 : ar);
  return x0 - tau * r;
}
template <class T> __host__ __device__ __forceinline__ T f1d_truncquad(T x0, T tau, T alpha, T beta) {                           // :277-291
  const T x_sq = f1d_square<T>(x0, 2 * tau * alpha);
  const T en_sq = alpha * x_sq * x_sq + (x_sq - x0) * (x_sq - x0) / (2 * tau);
  return en_sq < beta ? x_sq : x0;
}
template <class T> __host__ __device__ __forceinline__ T f1d_trunclin(T x0, T tau, T alpha, T beta) {                            // :309-323
  const T x_sh = f1d_abs<T>(x0, tau * alpha);
  const T en_sh = (x_sh - x0) * (x_sh - x0) / (2 * tau) + alpha * t_abs(x_sh);
  return en_sh < beta ? x_sh : x0;
}

}  // namespace elemop

// ---- the functors, names and call signature of the reference ----
#define PROST_FUNCTION_1D(NAME, EXPR)                                                             \
  template <typename T>                                                                           \
  struct NAME {                                                                                   \
    __host__ __device__ __forceinline__ T operator()(T x0, T tau, T alpha, T beta) const {        \
      (void)tau; (void)alpha; (void)beta;                                                         \
      return EXPR;                                                                                \
    }                                                                                             \
  };
PROST_FUNCTION_1D(Function1DZero, x0)
PROST_FUNCTION_1D(Function1DAbs, elemop::f1d_abs<T>(x0, tau))
PROST_FUNCTION_1D(Function1DSquare, elemop::f1d_square<T>(x0, tau))
PROST_FUNCTION_1D(Function1DIndLeq0, elemop::f1d_ind_leq0<T>(x0))
PROST_FUNCTION_1D(Function1DIndGeq0, elemop::f1d_ind_geq0<T>(x0))
PROST_FUNCTION_1D(Function1DIndEq0, (T)0)
PROST_FUNCTION_1D(Function1DIndBox01, elemop::f1d_ind_box01<T>(x0))
PROST_FUNCTION_1D(Function1DMaxPos0, elemop::f1d_max_pos0<T>(x0, tau))
PROST_FUNCTION_1D(Function1DL0, elemop::f1d_l0<T>(x0, tau))
PROST_FUNCTION_1D(Function1DHuber, elemop::f1d_huber<T>(x0, tau, alpha))
PROST_FUNCTION_1D(Function1DLq, elemop::f1d_lq<T>(x0, tau, alpha, beta))
PROST_FUNCTION_1D(Function1DLqPlusEps, (T)0)      /* a stub in the reference as well (:294-306) */
PROST_FUNCTION_1D(Function1DTruncQuad, elemop::f1d_truncquad<T>(x0, tau, alpha, beta))
PROST_FUNCTION_1D(Function1DTruncLinear, elemop::f1d_trunclin<T>(x0, tau, alpha, beta))
#undef PROST_FUNCTION_1D

}  // namespace prost
#endif
