// prost/prox/vector.hpp -- the view an elementwise operation gets of ONE element group of a variable.
//
// Plugin contract of the reference's include/prost/prox/vector.hpp:32-63: an ELEM_OPERATION functor receives
// `Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& tau_diag` and indexes components with
// operator[](i); component i of group tx sits at  interleaved ? tx * dim + i : tx + count * i  (vector.hpp:44-48).
//
// Device header (hipcc).  The generic kernel of prox_elem_operation.inl builds these views over HBM, over a
// workgroup's LDS tile or over a lane's private register tile (count = elements per lane) -- the index rule is the
// same in all three, and with a compile-time `dim` every index folds to a constant, i.e. a register.
#ifndef PROST_PROX_VECTOR_HPP_
#define PROST_PROX_VECTOR_HPP_
#include <hip/hip_runtime.h>

#include <cstddef>

namespace prost {

template <typename T>
class Vector {
 public:
  __host__ __device__ Vector(size_t count, size_t dim, bool interleaved, size_t tx, T* data)
      : data_(data), base_(interleaved ? tx * dim : tx), stride_(interleaved ? 1 : count) {}

  /// component i of this element group
  __host__ __device__ __forceinline__ T& operator[](size_t i) const { return data_[base_ + stride_ * i]; }

 private:
  T* data_;
  size_t base_, stride_;   // the two layouts differ only in where a group starts and how far its components lie apart
};

}  // namespace prost
#endif
