// prost/prox/elemop/elem_operation_norm2.hpp -- prox of  c f(a |x|_2 - b) + d |x|_2 + e/2 |x|_2^2  per element group.
//
// Plugin contract of the reference's include/prost/prox/elemop/elem_operation_norm2.hpp:36-98:
// ElemOperationNorm2<T, FUN_1D> : ElemOperation<0, 7> (dim is a run-time argument).  The squared norm is accumulated in
// component order, the scalar prox of elem_operation_1d.hpp runs on the norm, the step size is taken from tau_diag[0]
// only (:61 -- valid because diagsteps = false makes the preconditioner constant over a group, problem.cu:503-536), and
// a zero norm gives a zero result whatever the coefficients say (:84-90).
#ifndef PROST_PROX_ELEMOP_ELEM_OPERATION_NORM2_HPP_
#define PROST_PROX_ELEMOP_ELEM_OPERATION_NORM2_HPP_
#include "prost/prox/elemop/elem_operation_1d.hpp"

namespace prost {

template <typename T, class FUN_1D>
struct ElemOperationNorm2 : public ElemOperation<0, 7> {
  static const bool kWritesAllComponents = true;      // every res[i] is assigned on every path: the tile path need not preload res
  __host__ __device__ ElemOperationNorm2(T* coeffs, size_t dim, SharedMem<SharedMemType, GetSharedMemCount>& /*shared_mem*/)
      : coeffs_(coeffs), dim_(dim) {}

  __host__ __device__ __forceinline__ void operator()(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& tau_diag, T tau_scal,
                                                      bool invert_tau) {
    T norm = 0;
    for (size_t i = 0; i < dim_; i++) {
      const T val = arg[i];
      norm += val * val;
    }
    if (norm > 0) {
      norm = elemop::t_sqrt(norm);
      const T tau = elemop::StepSize(tau_scal, tau_diag[0], invert_tau);
      const T prox_result = elemop::ScaledProx<T, FUN_1D>(norm, tau, coeffs_);
      for (size_t i = 0; i < dim_; i++) res[i] = prox_result * arg[i] / norm;
    } else {
      for (size_t i = 0; i < dim_; i++) res[i] = 0;
    }
  }

 private:
  T* coeffs_;
  size_t dim_;
};

}  // namespace prost
#endif
