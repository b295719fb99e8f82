// prost/compat/thrust_ranges.hpp -- the reference's ITERATOR-RANGE virtuals on top of this library's pointer-range virtuals.
//
// The reference's plugin contract hands a block / prox its operands as thrust::device_vector<T>::iterator ranges
// (include/prost/linop/block.hpp:66-77, include/prost/prox/prox.hpp:117-126, prox_separable_sum.hpp); this library hands over raw
// HBM pointers (the host library is built by plain g++ and never sees a device header).  A block or prox
// written for the reference -- thrust algorithms over the ranges, or raw_pointer_cast(&(*it)) into its own kernels -- keeps its
// EvalLocalAdd / EvalAdjointLocalAdd / EvalLocal bodies UNCHANGED by deriving from the mix-ins below instead of prost::Block /
// prost::Prox / prost::ProxSeparableSum (or, in a source that says `namespace prost { ... }`, by `using namespace prost::thrust_api;`):
// the mix-in implements the pointer virtual once, wraps the pointers into the very iterator type the reference uses
// (thrust::device_vector<T>::iterator = normal_iterator<device_ptr<T>>: no copy, no allocation) and calls the iterator virtual.
// Compile the plugin with hipcc (rocThrust's HIP backend; /opt/rocm/include).  thrust algorithms called without an execution
// policy run on the NULL stream, which is the stream this library uses unless prost::SetCurrentStream was called; a plugin that
// must follow another stream passes thrust::hip::par.on((hipStream_t)prost::CurrentStream()).
// Checked by tests/plugins/thrust_style_plugins.hip (a block and a prox written with the reference's signatures, evaluated
// against the oracle): tests/test_plugins.py.
#ifndef PROST_COMPAT_THRUST_RANGES_HPP_
#define PROST_COMPAT_THRUST_RANGES_HPP_
#include <thrust/device_ptr.h>
#include <thrust/device_vector.h>

#include "prost/linop/block.hpp"
#include "prost/prox/prox.hpp"
#include "prost/prox/prox_separable_sum.hpp"

namespace prost {
namespace compat {

template <typename T> using device_iterator = typename thrust::device_vector<T>::iterator;
template <typename T> using device_const_iterator = typename thrust::device_vector<T>::const_iterator;
template <typename T> inline device_iterator<T> iter(T* p) { return device_iterator<T>(thrust::device_ptr<T>(p)); }
template <typename T> inline device_const_iterator<T> citer(const T* p) { return device_const_iterator<T>(thrust::device_ptr<const T>(p)); }

/// block.hpp:37-83 of the reference: override the two iterator-range virtuals, row_sum, col_sum, gpu_mem_amount
template <typename T>
class Block : public prost::Block<T> {
 public:
  Block(size_t row, size_t col, size_t nrows, size_t ncols) : prost::Block<T>(row, col, nrows, ncols) {}

 protected:
  virtual void EvalLocalAdd(const device_iterator<T>& res_begin, const device_iterator<T>& res_end, const device_const_iterator<T>& rhs_begin,
                            const device_const_iterator<T>& rhs_end) = 0;
  virtual void EvalAdjointLocalAdd(const device_iterator<T>& res_begin, const device_iterator<T>& res_end, const device_const_iterator<T>& rhs_begin,
                                   const device_const_iterator<T>& rhs_end) = 0;
  void EvalLocalAdd(T* res_begin, T* res_end, const T* rhs_begin, const T* rhs_end) final {
    EvalLocalAdd(iter(res_begin), iter(res_end), citer(rhs_begin), citer(rhs_end));
  }
  void EvalAdjointLocalAdd(T* res_begin, T* res_end, const T* rhs_begin, const T* rhs_end) final {
    EvalAdjointLocalAdd(iter(res_begin), iter(res_end), citer(rhs_begin), citer(rhs_end));
  }
};

/// the iterator-range EvalLocal of prox.hpp:117-126 on top of any of this library's prox bases
template <typename T, class BASE>
class ProxRanges : public BASE {
 public:
  using BASE::BASE;

 protected:
  virtual void EvalLocal(const device_iterator<T>& result_beg, const device_iterator<T>& result_end, const device_const_iterator<T>& arg_beg,
                         const device_const_iterator<T>& arg_end, const device_const_iterator<T>& tau_beg, const device_const_iterator<T>& tau_end,
                         T tau, bool invert_tau) = 0;
  void EvalLocal(T* result_beg, T* result_end, const T* arg_beg, const T* arg_end, const T* tau_beg, const T* tau_end, T tau, bool invert_tau) final {
    EvalLocal(iter(result_beg), iter(result_end), citer(arg_beg), citer(arg_end), citer(tau_beg), citer(tau_end), tau, invert_tau);
  }
};
template <typename T> using Prox = ProxRanges<T, prost::Prox<T>>;
template <typename T> using ProxSeparableSum = ProxRanges<T, prost::ProxSeparableSum<T>>;

}  // namespace compat

/// `using namespace prost::thrust_api;` inside a source written for the reference makes Block<T> / Prox<T> / ProxSeparableSum<T> name the
/// iterator-range classes (a using-directive does not hide prost::Block: qualify, or derive from prost::compat::Block<T> directly,
/// where both are visible)
namespace thrust_api {
using compat::Block;
using compat::Prox;
using compat::ProxSeparableSum;
}  // namespace thrust_api

}  // namespace prost
#endif
