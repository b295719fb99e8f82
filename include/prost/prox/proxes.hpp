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
/// (The reference's template itself -- ProxElemOperation<T, ELEM_OPERATION> for user-written
/// operations -- is prost/prox/prox_elem_operation.hpp.)
template <typename T>
class ProxElemDispatch : public ProxSeparableSum<T> {
 public:
  /// op: PROST_OP_1D (dim forced to 1, elem_operation_1d.hpp:30) or PROST_OP_NORM2
  ProxElemDispatch(int op, int fn, size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps,
                    const std::array<std::vector<T>, 7>& coeffs);
  virtual void Initialize();      // uploads per-element coefficient vectors (prox_elem_operation.inl:200-222)
  virtual void Release();
  virtual size_t gpu_mem_amount() const;
  virtual bool describe(ProxDesc& d) const;
  int op() const { return op_; }
  int fn() const { return fn_; }
  /// prox of the conjugate (Moreau) of this elem operation in one fused pass; same ranges as EvalLocal
  void EvalMoreauLocal(T* res, const T* arg, const T* tau_diag, T tau, bool invert_tau);
  virtual bool supports_arg_source() const { return true; }
  virtual bool takes_step_record() const { return true; }
  /// the operator sources run the 16-bytes-per-lane kernel only: planar layout (or the 1-D operation), range and count on 16-byte boundaries
  virtual bool supports_op_source() const;
  virtual void EvalFromSource(device_vector<T>& result, const typename Prox<T>::ArgSource& src, const device_vector<T>& tau_diag, T tau, bool invert_tau = false) {
    EvalSourceLocal(false, result, src, tau_diag, tau, invert_tau);
  }
  /// moreau = true: the conjugate's prox from the source (used by ProxMoreau::EvalFromSource)
  void EvalSourceLocal(bool moreau, device_vector<T>& result, const typename Prox<T>::ArgSource& src, const device_vector<T>& tau_diag, T tau, bool invert_tau);

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
  void CoeffArgs(const T* (&ptrs)[7], double (&vals)[7]) const;
  int op_, fn_;
  std::array<std::vector<T>, 7> coeffs_;
  std::array<device_vector<T>, 7> d_coeffs_;
};

/// prox of the conjugate via Moreau's identity (prox_moreau.cu:98-134)
template <typename T>
class ProxMoreau : public Prox<T> {
 public:
  /// MI355X addition: a conjugated elem operation runs as ONE kernel (default on)
  static void SetFuseElemOperations(bool on);
  explicit ProxMoreau(shared_ptr<Prox<T>> conjugate) : Prox<T>(*conjugate), conjugate_(conjugate) {}
  virtual void Initialize();
  virtual void Release();
  virtual size_t gpu_mem_amount() const { return this->size_ * sizeof(T) + conjugate_->gpu_mem_amount(); }
  virtual void get_separable_structure(std::vector<std::tuple<size_t, size_t, size_t>>& sep) { conjugate_->get_separable_structure(sep); }
  virtual bool supports_arg_source() const;
  virtual bool takes_step_record() const { return supports_arg_source(); }
  virtual bool supports_op_source() const { return supports_arg_source() && conjugate_->supports_op_source(); }
  /// the wrapped operation's description with `moreau` set (a wrap of a wrap is not described)
  virtual bool describe(ProxDesc& d) const;
  virtual void EvalFromSource(device_vector<T>& result, const typename Prox<T>::ArgSource& src, const device_vector<T>& tau_diag, T tau, bool invert_tau = false);
  virtual void average_preconditioner(std::vector<T>& precond) { conjugate_->average_preconditioner(precond); }
  virtual bool average_uniform(T& value) const { return conjugate_->average_uniform(value); }

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
  virtual bool supports_arg_source() const { return true; }
  virtual bool takes_step_record() const { return true; }
  virtual bool supports_op_source() const { const size_t v = 16 / sizeof(T); return this->size_ % v == 0 && this->index_ % v == 0; }
  /// the identity prox of a source IS the argument pass, written straight into the result
  virtual void EvalFromSource(device_vector<T>& result, const typename Prox<T>::ArgSource& src, const device_vector<T>& tau_diag, T tau, bool invert_tau = false);

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

/// elem_operation:ind_sum -- sum-to-one constraint per group (elem_operation_ind_sum.hpp:41-60)
template <typename T>
class ProxElemIndSum : public ProxSeparableSum<T> {
 public:
  ProxElemIndSum(size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps)
      : ProxSeparableSum<T>(index, count, dim, interleaved, diagsteps) {}
  virtual size_t gpu_mem_amount() const { return 0; }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
};

/// elem_operation:ind_simplex -- projection onto the unit simplex per group (elem_operation_ind_simplex.hpp:40-119)
template <typename T>
class ProxElemIndSimplex : public ProxSeparableSum<T> {
 public:
  ProxElemIndSimplex(size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps)
      : ProxSeparableSum<T>(index, count, dim, interleaved, diagsteps) {}
  virtual void Initialize() { work_.resize(this->count_ * this->dim_); }
  virtual void Release() { work_.clear(); }
  virtual size_t gpu_mem_amount() const { return this->count_ * this->dim_ * sizeof(T); }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
  device_vector<T> work_;      // sort workspace, planar like the data
};

/// h(x) = c f(ax - b) + dx + (e/2) x^2 around any inner prox (prox_transform.cu:99-221)
template <typename T>
class ProxTransform : public Prox<T> {
 public:
  ProxTransform(shared_ptr<Prox<T>> inner_fn, const std::vector<T>& a, const std::vector<T>& b, const std::vector<T>& c,
                const std::vector<T>& d, const std::vector<T>& e)
      : Prox<T>(*inner_fn), inner_fn_(inner_fn), host_{a, b, c, d, e} {}
  virtual void Initialize();
  virtual void Release();
  virtual size_t gpu_mem_amount() const;
  virtual void get_separable_structure(std::vector<std::tuple<size_t, size_t, size_t>>& sep) { inner_fn_->get_separable_structure(sep); }
  virtual void average_preconditioner(std::vector<T>& precond) { inner_fn_->average_preconditioner(precond); }
  virtual bool average_uniform(T& value) const { return inner_fn_->average_uniform(value); }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
  shared_ptr<Prox<T>> inner_fn_;
  std::array<std::vector<T>, 5> host_;
  std::array<device_vector<T>, 5> dev_;
  device_vector<T> scaled_arg_, scaled_tau_;
};

/// composition with a permutation of the (local) variable order (prox_permute.cu:50-160)
template <typename T>
class ProxPermute : public Prox<T> {
 public:
  ProxPermute(shared_ptr<Prox<T>> base_prox, const std::vector<int32_t>& perm) : Prox<T>(*base_prox), base_prox_(base_prox), perm_host_(perm) {}
  virtual void Initialize();
  virtual void Release();
  virtual size_t gpu_mem_amount() const { return this->size_ * sizeof(T) + base_prox_->gpu_mem_amount(); }
  virtual void get_separable_structure(std::vector<std::tuple<size_t, size_t, size_t>>& sep) { base_prox_->get_separable_structure(sep); }
  virtual void average_preconditioner(std::vector<T>& precond) { base_prox_->average_preconditioner(precond); }
  virtual bool average_uniform(T& value) const { return base_prox_->average_uniform(value); }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
  shared_ptr<Prox<T>> base_prox_;
  std::vector<int32_t> perm_host_;
  device_vector<int32_t> perm_;
  device_vector<T> permuted_arg_;
};

/// projection onto the halfspaces {x | <a, x> <= b} (prox_ind_halfspace.cu:31-153); planar layout always
template <typename T>
class ProxIndHalfspace : public ProxSeparableSum<T> {
 public:
  ProxIndHalfspace(size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps, const std::vector<T>& a, const std::vector<T>& b)
      : ProxSeparableSum<T>(index, count, dim, interleaved, diagsteps), a_(a), b_(b) {}
  virtual void Initialize();
  virtual void Release() { d_a_.clear(); d_b_.clear(); }
  virtual size_t gpu_mem_amount() const { return (a_.size() + b_.size()) * sizeof(T); }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
  std::vector<T> a_, b_;
  device_vector<T> d_a_, d_b_;
};

/// projection onto the second-order cone alpha ||x|| <= y, alpha == 1 only (prox_ind_soc.cu:30-124)
template <typename T>
class ProxIndSOC : public ProxSeparableSum<T> {
 public:
  ProxIndSOC(size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps, T alpha)
      : ProxSeparableSum<T>(index, count, dim, interleaved, diagsteps), alpha_(alpha) {}
  virtual void Initialize();
  virtual size_t gpu_mem_amount() const { return 0; }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
  T alpha_;
};

/// sum constraints over one or two index families, identity elsewhere (prox_ind_sum.cu:30-147)
template <typename T>
class ProxIndSum : public Prox<T> {
 public:
  ProxIndSum(size_t index, size_t size, size_t count, size_t dim, const std::vector<uint64_t>& inds, T sum)
      : Prox<T>(index, size, true), dim_(dim), dim_2_(0), count_(count), count_2_(0), inds_(inds), two_(false), sum_(sum), sum_2_(0) {}
  ProxIndSum(size_t index, size_t size, size_t count, size_t dim, const std::vector<uint64_t>& inds, T sum, size_t count2, size_t dim2,
             const std::vector<uint64_t>& inds2, T sum2)
      : Prox<T>(index, size, true), dim_(dim), dim_2_(dim2), count_(count), count_2_(count2), inds_(inds), inds_2_(inds2), two_(true),
        sum_(sum), sum_2_(sum2) {}
  virtual void Initialize();
  virtual void Release() { d_inds_.clear(); d_inds_2_.clear(); }
  virtual size_t gpu_mem_amount() const { return inds_.size() * sizeof(uint64_t); }

 protected:
  virtual void EvalLocal(T*, T*, const T*, const T*, const T*, const T*, T tau, bool invert_tau);
  size_t dim_, dim_2_, count_, count_2_;
  std::vector<uint64_t> inds_, inds_2_;
  device_vector<int64_t> d_inds_, d_inds_2_;
  bool two_;
  T sum_, sum_2_;
};

}  // namespace prost
#endif
