// prost/problem.hpp -- min_x max_y g(x) + <Kx,y> - f*(y): linear operator, prox lists and the
// diagonal preconditioners (reference include/prost/problem.hpp:46-158, src/problem.cu).
#ifndef PROST_PROBLEM_HPP_
#define PROST_PROBLEM_HPP_
#include "prost/linop/linearoperator.hpp"
#include "prost/prox/prox.hpp"

namespace prost {

template <typename T>
class Problem {
 public:
  enum Scaling { kScalingIdentity, kScalingAlpha, kScalingCustom };
  typedef std::vector<shared_ptr<Prox<T>>> ProxList;

  Problem();
  virtual ~Problem() {}

  void AddBlock(shared_ptr<Block<T>> block);
  void AddProx_g(shared_ptr<Prox<T>> prox) { prox_g_.push_back(prox); }
  void AddProx_f(shared_ptr<Prox<T>> prox) { prox_f_.push_back(prox); }
  void AddProx_gstar(shared_ptr<Prox<T>> prox) { prox_gstar_.push_back(prox); }
  void AddProx_fstar(shared_ptr<Prox<T>> prox) { prox_fstar_.push_back(prox); }

  /// Pock-Chambolle: Sigma_i = 1 / sum_j |K_ij|^alpha, Tau_j = 1 / sum_i |K_ij|^(2-alpha)
  void SetScalingAlpha(T alpha) { scaling_type_ = kScalingAlpha; scaling_alpha_ = alpha; }
  void SetScalingIdentity() { scaling_type_ = kScalingIdentity; }
  /// user vectors are squared on entry (problem.cu:357-380)
  void SetScalingCustom(const std::vector<T>& left, const std::vector<T>& right);
  void SetDimensions(size_t nrows, size_t ncols) { nrows_ = nrows; ncols_ = ncols; }

  /// checks, zero-prox filling, preconditioners + averaging -- host only (problem.cu:196-300)
  void InitializeHost();
  /// InitializeHost + device uploads (problem.cu:196-323)
  void Initialize();
  void Release();
  /// swap primal and dual roles (problem.cu:539-547)
  void Dualize();
  /// power iteration estimate of |Sigma^(1/2) K Tau^(1/2)| (problem.cu:429-500)
  T normest(T tol = 1e-6, int max_iters = 100);

  shared_ptr<LinearOperator<T>> linop() const { return linop_; }
  const device_vector<T>& scaling_left() const { return scaling_left_; }
  const device_vector<T>& scaling_right() const { return scaling_right_; }
  /// host copies; when the preconditioner is one constant (uniform_left / uniform_right) they are only filled on first use
  const std::vector<T>& scaling_left_host() const { MaterializeHost(); return scaling_left_host_; }
  const std::vector<T>& scaling_right_host() const { MaterializeHost(); return scaling_right_host_; }
  /// MI355X addition: Sigma (left) / Tau (right) known to be ONE value (a single stencil block with constant row and
  /// column sums): nothing is swept, stored or uploaded per entry -- at 10^9 entries that is seconds of setup
  bool uniform_left(T& v) const { v = left_value_; return left_uniform_; }
  bool uniform_right(T& v) const { v = right_value_; return right_uniform_; }
  const ProxList& prox_f() const { return prox_f_; }
  const ProxList& prox_g() const { return prox_g_; }
  const ProxList& prox_fstar() const { return prox_fstar_; }
  const ProxList& prox_gstar() const { return prox_gstar_; }
  size_t nrows() const { return nrows_; }
  size_t ncols() const { return ncols_; }
  size_t gpu_mem_amount() const;

 protected:
  void AveragePreconditioners(std::vector<T>& precond, const ProxList& prox);
  void MaterializeHost() const;

  size_t nrows_, ncols_;
  shared_ptr<LinearOperator<T>> linop_, dual_linop_;
  Scaling scaling_type_;
  device_vector<T> scaling_left_, scaling_right_;           // squared preconditioners Sigma, Tau
  mutable std::vector<T> scaling_left_host_, scaling_right_host_;
  bool left_uniform_ = false, right_uniform_ = false;
  T left_value_ = 0, right_value_ = 0;
  T scaling_alpha_;
  ProxList prox_f_, prox_g_, prox_fstar_, prox_gstar_;
  bool host_initialized_;
};

}  // namespace prost
#endif
