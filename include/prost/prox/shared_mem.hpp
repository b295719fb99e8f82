// prost/prox/shared_mem.hpp -- per-lane scratch slice in LDS for elementwise operations that need one.
//
// Plugin contract of the reference's include/prost/prox/shared_mem.hpp:28-62: an ELEM_OPERATION names its scratch
// element type (`SharedMemType`) and a functor `GetSharedMemCount` (dim -> entries per thread); the launcher reserves
// count(dim) * blockDim.x * sizeof(SharedMemType) bytes of dynamic LDS and every thread indexes ITS entries with
// operator[](i), i < count(dim).
//
// gfx950 layout: entry i of lane t lives at  i * blockDim.x + t  -- the 64 lanes of a wavefront touch 64 consecutive
// words for any i, so an access is conflict-free over the 64 LDS banks (the reference lays the slices out
// thread-major, t * count + i, which serialises 2- to 32-way whenever count shares a factor with the bank count).
// The slice is private to a lane either way, so an operation cannot observe the difference.
#ifndef PROST_PROX_SHARED_MEM_HPP_
#define PROST_PROX_SHARED_MEM_HPP_
#include <hip/hip_runtime.h>

#include <cstddef>

namespace prost {

template <typename T, class F>
class SharedMem {
 public:
  /// dim: the operation's dimension (argument of F); threadIdx_x: this lane's index in its workgroup
  __device__ SharedMem(size_t dim, size_t threadIdx_x) : lane_(threadIdx_x), lanes_(blockDim.x) {
    extern __shared__ __attribute__((aligned(16))) char prost_elem_operation_lds[];
    slice_ = reinterpret_cast<T*>(prost_elem_operation_lds);
    (void)dim;
  }
  __device__ __forceinline__ T operator[](size_t i) const { return slice_[i * lanes_ + lane_]; }
  __device__ __forceinline__ T& operator[](size_t i) { return slice_[i * lanes_ + lane_]; }

 private:
  size_t lane_, lanes_;
  T* slice_;
};

}  // namespace prost
#endif
