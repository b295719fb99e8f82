// prost/prox/elemop/elem_operation_ind_simplex.hpp -- projection of every element group onto the unit simplex.
//
// Plugin contract of the reference's include/prost/prox/elemop/elem_operation_ind_simplex.hpp:40-119 (ElemOperation<0, 0, T>; the
// algorithm of arXiv:1101.6081: sort descending, find the threshold, clip).  The reference sorts a thread-local array of a fixed
// maximal length; here the group is sorted in the operation's per-lane LDS slice -- GetSharedMemCount(dim) = dim entries of T, the
// hook the reference's header provides for exactly this (shared_mem.hpp) -- with the same Shell sort gaps, so the sums that
// determine the threshold are formed in the same order and the result equals the built-in `elem_operation:ind_simplex` bit for bit.
#ifndef PROST_PROX_ELEMOP_ELEM_OPERATION_IND_SIMPLEX_HPP_
#define PROST_PROX_ELEMOP_ELEM_OPERATION_IND_SIMPLEX_HPP_
#include "prost/prox/elemop/elem_operation.hpp"

namespace prost {

template <typename T>
struct ElemOperationIndSimplex : public ElemOperation<0, 0, T> {
  static const bool kWritesAllComponents = true;      // every res[i] is assigned on every path: the tile path need not preload res
  struct GetSharedMemCount {
    __host__ __device__ size_t operator()(size_t dim) { return dim; }
  };
  __device__ ElemOperationIndSimplex(size_t dim, SharedMem<T, GetSharedMemCount>& shared_mem) : dim_(dim), sorted_(shared_mem) {}

  __device__ inline void operator()(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& /*tau_diag*/, T /*tau_scal*/, bool /*invert_tau*/) {
    for (size_t i = 0; i < dim_; i++) sorted_[i] = arg[i];
    const int gaps[6] = {132, 57, 23, 10, 4, 1};               // descending Shell sort (:98-115)
    for (int k = 0; k < 6; k++) {
      const int gap = gaps[k];
      for (int i = gap; i < (int)dim_; i++) {
        const T temp = sorted_[i];
        int j = i;
        for (; (j >= gap) && (sorted_[j - gap] <= temp); j -= gap) sorted_[j] = sorted_[j - gap];
        sorted_[j] = temp;
      }
    }
    bool found = false;                                          // :72-87
    T tmpsum = 0, tmax = 0;
    for (int ii = 1; ii <= (int)dim_ - 1; ii++) {
      tmpsum += sorted_[ii - 1];
      tmax = (T)(((double)tmpsum - 1.) / (double)(T)ii);
      if (tmax >= sorted_[ii]) { found = true; break; }
    }
    if (!found) tmax = (T)(((double)(tmpsum + sorted_[dim_ - 1]) - 1.0) / (double)(T)dim_);
    for (size_t i = 0; i < dim_; i++) {
      const T v = arg[i] - tmax;
      res[i] = v > (T)0 ? v : (T)0;
    }
  }

 private:
  size_t dim_;
  SharedMem<T, GetSharedMemCount>& sorted_;
};

}  // namespace prost
#endif
