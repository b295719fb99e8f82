// prost/prox/prox.hpp -- plugin base class of proximal operators.
//
// Same contract as the reference's include/prost/prox/prox.hpp:39-135: a prox covers
// [index, index+size) of its variable; Eval slices result / arg / tau_diag and calls EvalLocal.
// Ranges are raw HBM pointers instead of thrust iterators; kernels go to prost::CurrentStream()
// and there is NO device synchronisation after a prox (the reference synchronises after every
// launch, prox_elem_operation.inl:128,187).
#ifndef PROST_PROX_PROX_HPP_
#define PROST_PROX_PROX_HPP_
#include "prost_hip.h"
#include "prost/common.hpp"
#include "prost/device_vector.hpp"

namespace prost {

template <typename T> class ProxMoreau;

/// what a backend needs to fuse an elementwise prox into its passes
struct ProxDesc {
  enum Kind { kNone = 0, kElem1D, kElemNorm2 } kind = kNone;
  int fn = 0;
  size_t count = 0, dim = 0;
  bool interleaved = false;
  const void* coeff_ptr[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // device, or null
  double coeff_val[7] = {0, 0, 0, 0, 0, 0, 0};
  bool moreau = false;      // the prox is the Moreau wrap (prox_moreau.cu:98-134) of the described operation
};

template <typename T> class ProxTransform;
template <typename T> class ProxPermute;

template <typename T>
class Prox {
  friend class ProxMoreau<T>;
  friend class ProxTransform<T>;
  friend class ProxPermute<T>;

 public:
  Prox(size_t index, size_t size, bool diagsteps) : index_(index), size_(size), diagsteps_(diagsteps) {}
  Prox(const Prox<T>& other) : index_(other.index_), size_(other.size_), diagsteps_(other.diagsteps_) {}
  virtual ~Prox() {}

  virtual void Initialize() {}
  virtual void Release() {}

  /// result[index:index+size] = prox(arg[index:...]; tau * tau_diag[index:...])   (prox.cu:27-43)
  void Eval(device_vector<T>& result, const device_vector<T>& arg, const device_vector<T>& tau_diag, T tau,
            bool invert_tau = false);
  /// host-vector version for debugging / eval_prox; returns milliseconds (prox.cu:46-71)
  double Eval(std::vector<T>& result, const std::vector<T>& arg, const std::vector<T>& tau_diag, T tau);

  virtual size_t gpu_mem_amount() const = 0;
  size_t index() const { return index_; }
  size_t size() const { return size_; }
  size_t end() const { return index_ + size_ - 1; }
  bool diagsteps() const { return diagsteps_; }

  /// (start, count, stride) groups over which a preconditioner must be constant (prox.cu:74-78)
  virtual void get_separable_structure(std::vector<std::tuple<size_t, size_t, size_t>>& sep) {
    sep.push_back(std::tuple<size_t, size_t, size_t>(index_, size_, 1));
  }
  /// Replaces the preconditioner entries of each separable group by the group mean (problem.cu:503-536).
  /// Default: over the groups of get_separable_structure; proxes with a regular structure override
  /// this with streaming loops of the same arithmetic (no per-group tuple at 10^7 groups).
  virtual void average_preconditioner(std::vector<T>& precond) {
    std::vector<std::tuple<size_t, size_t, size_t>> groups;
    get_separable_structure(groups);
    for (auto& g : groups) {
      const size_t idx = std::get<0>(g), cnt = std::get<1>(g), stride = std::get<2>(g);
      T avg = 0;
      for (size_t c = 0; c < cnt; c++) avg += precond[idx + c * stride];
      avg /= static_cast<T>(cnt);
      for (size_t c = 0; c < cnt; c++) precond[idx + c * stride] = avg;
    }
  }
  /// MI355X addition: the preconditioner over this prox's range is the ONE value `value`; if averaging leaves it one value,
  /// store it (the mean as average_preconditioner forms it) and return true.  false = unknown: the caller materialises the
  /// vector and calls average_preconditioner.
  virtual bool average_uniform(T& value) const { (void)value; return false; }
  virtual bool describe(ProxDesc&) const { return false; }

  /// A prox argument that is not materialised: the backend names the vectors and scalars it is made of
  /// (prost_hip_arg_spec modes; pointers to element 0 of the whole variable) and a prox that can form it on the
  /// fly evaluates straight from them -- the separate argument pass of the PDHG iteration disappears.
  struct ArgSource {
    int mode;               ///< PROST_ARG_PDHG_PRIMAL (x, T, K^T y; tau) or PROST_ARG_PDHG_DUAL (y, Sigma, K x, K x_prev; sigma, theta)
    const T* v[4];
    T s[2];
    // PROST_ARG_PDHG_PRIMAL_OP / _DUAL_OP (round 5): the operator product is formed on the fly as well (prost_hip_arg_spec) -- v[2] (and v[3]
    // of the dual source) are then unused; v[3] of the primal source = K^T y_prev (read for the residual sums)
    const prost_hip_fused_op* op = nullptr;
    size_t op_rows = 0, op_cols = 0;
    const T* w[2] = {nullptr, nullptr};     ///< whole vectors the product is taken of: y | x, x_prev
    T* kty_out = nullptr;                   ///< primal source: K^T y is stored here (element 0 of the whole variable)
    int use[2] = {1, 1};                    ///< 0: that product counts as the zero vector (iterations 0 / 1 of the reference)
    double* res_ws = nullptr;               ///< residual sums: slots of 4 doubles; *res_slot is advanced by the slots a launch takes
    unsigned* res_slot = nullptr;
    unsigned res_slots_max = 0;             ///< slots ONE launch may take
  };
  /// true if EvalFromSource also takes the operator sources (PROST_ARG_PDHG_PRIMAL_OP / _DUAL_OP)
  virtual bool supports_op_source() const { return false; }
  /// true if EvalFromSource is implemented; a backend only skips its argument pass when EVERY prox of the list is
  virtual bool supports_arg_source() const { return false; }
  /// EvalFromSource launches only kernels that take their step sizes from the device-resident record of a batch of iterations
  /// (prost_hip_use_step_record): the in-tree elem operations, their Moreau wraps and the identity.  A prox of a plugin that
  /// implements EvalFromSource with kernels of its own keeps the default -- its problems run the host loop.
  virtual bool takes_step_record() const { return false; }
  /// MI355X addition (round 5): the step size as a DEVICE scalar.  Inside a batch of iterations whose step-size rule runs on the device
  /// (goldstein / boyd without a host wait per iteration) the host does not know tau when it enqueues the prox.  A prox whose kernels can
  /// read it from device memory returns true from takes_step_view() and implements EvalLocalStepView: `*view.step` is the scalar the
  /// reference would have passed as `tau`, valid when the kernel RUNS; a non-zero `*view.stop` means the batch has ended (the stopping
  /// test fired) and the kernel must leave `result` untouched.  ProxElemOperation<T, OP> (prox_elem_operation.inl) implements it for every
  /// user-written operation; proxes that keep the default run the host loop, as the reference does.
  struct StepView { const T* step; const int* stop; };
  virtual bool takes_step_view() const { return false; }
  void EvalWithStepView(device_vector<T>& result, const device_vector<T>& arg, const device_vector<T>& tau_diag, const StepView& view, bool invert_tau = false);
  /// result[index:index+size) = prox(source[index:index+size); tau * tau_diag[...]); result must not alias source.v[0]
  virtual void EvalFromSource(device_vector<T>& result, const ArgSource& src, const device_vector<T>& tau_diag, T tau, bool invert_tau = false) {
    (void)result; (void)src; (void)tau_diag; (void)tau; (void)invert_tau;
    throw Exception("This prox cannot evaluate from an argument source.");
  }

 protected:
  virtual void EvalLocal(T* result_beg, T* result_end, const T* arg_beg, const T* arg_end, const T* tau_beg,
                         const T* tau_end, T tau, bool invert_tau) = 0;
  /// EvalLocal with the step size read on the device (StepView above); the default refuses
  virtual void EvalLocalStepView(T* result_beg, T* result_end, const T* arg_beg, const T* arg_end, const T* tau_beg, const T* tau_end,
                                 const StepView& view, bool invert_tau) {
    (void)result_beg; (void)result_end; (void)arg_beg; (void)arg_end; (void)tau_beg; (void)tau_end; (void)view; (void)invert_tau;
    throw Exception("This prox cannot take its step size from the device.");
  }
  size_t index_, size_;
  bool diagsteps_;
};

}  // namespace prost
#endif
