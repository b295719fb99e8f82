// prost/prox/elemop/elem_operation_1d.hpp -- prox of  c f(a x - b) + d x + e/2 x^2  per element, f a Function1D.
//
// Plugin contract of the reference's include/prost/prox/elemop/elem_operation_1d.hpp:30-64:
// ElemOperation1D<T, FUN_1D> : ElemOperation<1, 7>, coefficients (a, b, c, d, e, alpha, beta).  Compose it with one of
// function_1d.hpp's functors or with your own `struct MyFun { T operator()(T x0, T tau, T alpha, T beta) const; }`:
//     ProxElemOperation<T, ElemOperation1D<T, MyFun<T>>>(idx, count, 1, false, diagsteps, coeffs)
// Expression order and the double-literal promotions follow the reference line by line (the built-in
// elem_operation:1d:* operations, which dispatch the same arithmetic at run time, agree bit for bit; tests).
#ifndef PROST_PROX_ELEMOP_ELEM_OPERATION_1D_HPP_
#define PROST_PROX_ELEMOP_ELEM_OPERATION_1D_HPP_
#include "prost/prox/elemop/elem_operation.hpp"
#include "prost/prox/elemop/function_1d.hpp"

namespace prost {
namespace elemop {

/// step size of one element group: invert_tau ? 1 / (tau_scal * tau_diag) : tau_scal * tau_diag   (elem_operation_1d.hpp:40)
template <class T>
__host__ __device__ __forceinline__ T StepSize(T tau_scal, T tau_diag, bool invert_tau) {
  return invert_tau ? (T)(1. / (double)(tau_scal * tau_diag)) : (tau_scal * tau_diag);
}
/// the scaled scalar prox of elem_operation_1d.hpp:45-57 (== elem_operation_norm2.hpp:64-74) applied to the value v
template <class T, class FUN_1D>
__host__ __device__ __forceinline__ T ScaledProx(T v, T tau, const T* c) {
  const double den = 1. + (double)(tau * c[4]);
  const T prox_arg = (T)(div1((double)(c[0] * (v - c[3] * tau)), den) - (double)c[1]);
  const T step = (T)div1((double)(c[2] * c[0] * c[0] * tau), den);
  FUN_1D fun;
  return div1((T)(fun(prox_arg, step, c[5], c[6]) + c[1]), c[0]);
}

}  // namespace elemop

template <typename T, class FUN_1D>
struct ElemOperation1D : public ElemOperation<1, 7> {
  static const bool kWritesAllComponents = true;      // every res[i] is assigned on every path: the tile path need not preload res
  __host__ __device__ ElemOperation1D(T* coeffs, size_t /*dim*/, SharedMem<SharedMemType, GetSharedMemCount>& /*shared_mem*/) : coeffs_(coeffs) {}

  __host__ __device__ __forceinline__ void operator()(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& tau_diag, T tau_scal,
                                                      bool invert_tau) {
    const T tau = elemop::StepSize(tau_scal, tau_diag[0], invert_tau);
    if (coeffs_[0] == 0 || coeffs_[2] == 0) res[0] = (arg[0] - tau * coeffs_[3]) / (1 + tau * coeffs_[4]);      // :42-44: f drops out
    else res[0] = elemop::ScaledProx<T, FUN_1D>(arg[0], tau, coeffs_);
  }

 private:
  T* coeffs_;
};

}  // namespace prost
#endif
