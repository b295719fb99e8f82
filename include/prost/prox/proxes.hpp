// prost/prox/proxes.hpp -- in-tree proximal operators of the hot path (reference:
// prox_elem_operation.hpp/.inl + elemop/*, prox_moreau.hpp, prox_zero.hpp, prox_ind_epi_quad.hpp).
#ifndef PROST_PROX_PROXES_HPP_
#define PROST_PROX_PROXES_HPP_
#include <array>

#include "prost/prox/prox_separable_sum.hpp"

namespace prost {

/// The reference instantiates ProxElemOperation<T, ElemOperation1D|Norm2<T, Function1D*>> 176
/// times (prox_elem_operation.cu:44-256); here the elem operation and the scalar function are
/// run-time ids (PROST_OP_*, PROST_FN_* of prost_hip.h) dispatched wave-uniformly in one kernel.
template <typename T>
class ProxElemOperation : public ProxSeparableSum<T> {
 public:
  /// op: PROST_OP_1D (dim forced to 1, elem_operation_1d.hpp:30) or PROST_OP_NORM2
  ProxElemOperation(int op, int fn, size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps,
                    const std::array<std::vector<T>, 7>& coeffs);
  virtual void Initialize();      // uploads per-element coefficient vectors (prox_elem_operation.inl:200-222)
  virtual void Release();
  virtual size_t gpu_mem_amount() const;
  virtual bool describe(ProxDesc& d) const;
  int op() const { return op_; }
  int fn() const { return fn_; }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
  int op_, fn_;
  std::array<std::vector<T>, 7> coeffs_;
  std::array<device_vector<T>, 7> d_coeffs_;
};

/// prox of the conjugate via Moreau's identity (prox_moreau.cu:98-134)
template <typename T>
class ProxMoreau : public Prox<T> {
 public:
  explicit ProxMoreau(shared_ptr<Prox<T>> conjugate) : Prox<T>(*conjugate), conjugate_(conjugate) {}
  virtual void Initialize();
  virtual void Release();
  virtual size_t gpu_mem_amount() const { return this->size_ * sizeof(T) + conjugate_->gpu_mem_amount(); }
  virtual void get_separable_structure(std::vector<std::tuple<size_t, size_t, size_t>>& sep) { conjugate_->get_separable_structure(sep); }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
  shared_ptr<Prox<T>> conjugate_;
  device_vector<T> scaled_arg_;
};

/// identity (prox_zero.cu:37-48)
template <typename T>
class ProxZero : public Prox<T> {
 public:
  ProxZero(size_t index, size_t size) : Prox<T>(index, size, true) {}
  virtual size_t gpu_mem_amount() const { return 0; }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
};

/// projection onto the epigraph of a |x|^2 + b^T x + c (prox_ind_epi_quad.cu:42-169)
template <typename T>
class ProxIndEpiQuad : public ProxSeparableSum<T> {
 public:
  ProxIndEpiQuad(size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps, const std::vector<T>& a,
                 const std::vector<T>& b, const std::vector<T>& c)
      : ProxSeparableSum<T>(index, count, dim, interleaved, diagsteps), a_(a), b_(b), c_(c) {}
  virtual void Initialize();
  virtual void Release();
  virtual size_t gpu_mem_amount() const { return (a_.size() + b_.size() + c_.size()) * sizeof(T); }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
  std::vector<T> a_, b_, c_;
  device_vector<T> d_a_, d_b_, d_c_;
};

}  // namespace prost
#endif
