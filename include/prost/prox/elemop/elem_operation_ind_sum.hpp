// prost/prox/elemop/elem_operation_ind_sum.hpp -- projection of every element group onto { x : sum_i x_i = 1 }.
//
// Plugin contract of the reference's include/prost/prox/elemop/elem_operation_ind_sum.hpp:31-62 (ElemOperation<0, 0, T>: run-time
// dim, no coefficients): res = arg - (sum(arg) - 1) / dim, the sum taken in component order, the "1." promoting to double as in the
// reference.  The built-in `elem_operation:ind_sum` runs the same arithmetic (tests: bit for bit).
#ifndef PROST_PROX_ELEMOP_ELEM_OPERATION_IND_SUM_HPP_
#define PROST_PROX_ELEMOP_ELEM_OPERATION_IND_SUM_HPP_
#include "prost/prox/elemop/elem_operation.hpp"

namespace prost {

template <typename T>
struct ElemOperationIndSum : public ElemOperation<0, 0, T> {
  static const bool kWritesAllComponents = true;      // every res[i] is assigned on every path: the tile path need not preload res
  __host__ __device__ ElemOperationIndSum(size_t dim, SharedMem<typename ElemOperationIndSum::SharedMemType, typename ElemOperationIndSum::GetSharedMemCount>& /*shared_mem*/)
      : dim_(dim) {}

  __host__ __device__ __forceinline__ void operator()(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& /*tau_diag*/, T /*tau_scal*/,
                                                      bool /*invert_tau*/) {
    T tl = 0;
    for (size_t i = 0; i < dim_; i++) tl += arg[i];
    tl = (T)(((double)tl - 1.) / (double)static_cast<T>(dim_));
    for (size_t i = 0; i < dim_; i++) res[i] = arg[i] - tl;
  }

 private:
  size_t dim_;
};

}  // namespace prost
#endif
