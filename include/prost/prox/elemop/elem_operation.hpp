// prost/prox/elemop/elem_operation.hpp -- base of the user-written elementwise operations.
//
// Plugin contract of the reference's include/prost/prox/elemop/elem_operation.hpp:30-40.  An operation is a struct
//
//   template <typename T> struct MyOp : prost::ElemOperation<DIM, COEFFS_COUNT[, SHARED_MEM_TYPE]> {
//     __host__ __device__ MyOp(T* coeffs, size_t dim, prost::SharedMem<SharedMemType, GetSharedMemCount>& sh);   // COEFFS_COUNT != 0
//     __host__ __device__ MyOp(size_t dim, prost::SharedMem<SharedMemType, GetSharedMemCount>& sh);              // COEFFS_COUNT == 0
//     __host__ __device__ void operator()(prost::Vector<T>& res, const prost::Vector<const T>& arg,
//                                         const prost::Vector<const T>& tau_diag, T tau_scal, bool invert_tau);
//   };
//
// DIM = 0: the dimension of an element group is a run-time argument of the prox; DIM > 0 fixes it
// (prox_elem_operation.hpp:47, :76).  An operation that needs per-thread scratch shadows GetSharedMemCount
// (entries per thread as a function of dim; elem_operation_ind_simplex.hpp:44-46 is the reference's example).
// ProxElemOperation<T, MyOp<T>> (prost/prox/prox_elem_operation.hpp) turns it into a prost::Prox.
#ifndef PROST_PROX_ELEMOP_ELEM_OPERATION_HPP_
#define PROST_PROX_ELEMOP_ELEM_OPERATION_HPP_
#include "prost/prox/shared_mem.hpp"
#include "prost/prox/vector.hpp"

namespace prost {

template <size_t DIM = 0, size_t COEFFS_COUNT = 0, typename SHARED_MEM_TYPE = char>
struct ElemOperation {
  static const size_t kCoeffsCount = COEFFS_COUNT;
  static const size_t kDim = DIM;
  typedef SHARED_MEM_TYPE SharedMemType;
  struct GetSharedMemCount {
    __host__ __device__ size_t operator()(size_t /*dim*/) { return 0; }
  };
  /// MI355X addition.  The generic kernel keeps the element groups of a lane in registers when it can (16-byte
  /// accesses, prox_elem_operation.inl) and then stores every component of `res`; an operation that leaves some
  /// res[i] unwritten on purpose (so that the old content survives) shadows this with `true` and runs on the
  /// one-group-per-lane path over HBM, whose stores are exactly the operation's own.
  static const bool kPartialResult = false;
  /// MI355X addition, the opt-in counterpart.  By default the register-tile path first loads the old content of `res` into
  /// the tile, so an operation ported from the reference that skips some res[i] without saying so still behaves as it
  /// does there (the reference's kernel writes through views over global memory: what is not written stays).  An
  /// operation that assigns EVERY component of `res` on every path shadows this with `true` and saves that read.
  static const bool kWritesAllComponents = false;
};

}  // namespace prost
#endif
