// prox.cpp -- proximal operators of the prost host library (calls prost_hip.h only).
#include <algorithm>
#include <chrono>
#include <sstream>

#include "hipapi.hpp"
#include "prost/prox/proxes.hpp"

namespace prost {

template <typename T>
void Prox<T>::Eval(device_vector<T>& result, const device_vector<T>& arg, const device_vector<T>& tau_diag, T tau, bool invert_tau) {
  EvalLocal(result.data() + index_, result.data() + index_ + size_, arg.data() + index_, arg.data() + index_ + size_,
            tau_diag.data() + index_, tau_diag.data() + index_ + size_, tau, invert_tau);
}
template <typename T>
void Prox<T>::EvalWithStepView(device_vector<T>& result, const device_vector<T>& arg, const device_vector<T>& tau_diag, const StepView& view, bool invert_tau) {
  EvalLocalStepView(result.data() + index_, result.data() + index_ + size_, arg.data() + index_, arg.data() + index_ + size_,
                    tau_diag.data() + index_, tau_diag.data() + index_ + size_, view, invert_tau);
}
template <typename T>
double Prox<T>::Eval(std::vector<T>& result, const std::vector<T>& arg, const std::vector<T>& tau_diag, T tau) {
  device_vector<T> d_arg; d_arg = arg;
  device_vector<T> d_tau; d_tau = tau_diag;
  device_vector<T> d_res(arg.size());
  CheckHip(prost_hip_stream_synchronize(CurrentStream()), "sync");
  const auto t0 = std::chrono::steady_clock::now();
  Eval(d_res, d_arg, d_tau, tau);
  CheckHip(prost_hip_stream_synchronize(CurrentStream()), "sync");
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  CheckHip(prost_hip_check_last_error(), "prox kernel");
  d_res.copy_to(result);
  return ms;
}
template class Prox<float>;
template class Prox<double>;

// ---- elementwise 1d / norm2 ----
template <typename T>
ProxElemDispatch<T>::ProxElemDispatch(int op, int fn, size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps,
                                        const std::array<std::vector<T>, 7>& coeffs)
    : ProxSeparableSum<T>(index, count, op == PROST_OP_1D ? 1 : dim, interleaved, diagsteps), op_(op), fn_(fn), coeffs_(coeffs) {
  for (int i = 0; i < 7; i++)
    if (coeffs_[i].empty()) throw Exception("Empty vector passed.");
}
template <typename T>
void ProxElemDispatch<T>::Initialize() {
  for (int i = 0; i < 7; i++)
    if (coeffs_[i].size() > 1) {
      if (coeffs_[i].size() < this->count_) throw Exception("Size of coefficients should be either 1 or count.");
      d_coeffs_[i] = coeffs_[i];
    }
}
template <typename T> void ProxElemDispatch<T>::Release() { for (int i = 0; i < 7; i++) d_coeffs_[i].clear(); }
template <typename T>
size_t ProxElemDispatch<T>::gpu_mem_amount() const {
  size_t mem = 0;
  for (int i = 0; i < 7; i++) if (coeffs_[i].size() > 1) mem += this->count_ * sizeof(T);
  return mem;
}
template <typename T>
bool ProxElemDispatch<T>::describe(ProxDesc& d) const {
  d.kind = op_ == PROST_OP_1D ? ProxDesc::kElem1D : ProxDesc::kElemNorm2;
  d.fn = fn_; d.count = this->count_; d.dim = this->dim_; d.interleaved = this->interleaved_;
  for (int i = 0; i < 7; i++) {
    if (coeffs_[i].size() > 1) {
      if (d_coeffs_[i].size() != coeffs_[i].size()) return false;     // not initialised yet
      d.coeff_ptr[i] = d_coeffs_[i].data(); d.coeff_val[i] = 0;
    } else { d.coeff_ptr[i] = nullptr; d.coeff_val[i] = (double)coeffs_[i][0]; }
  }
  return true;
}
/// device pointer (per-element coefficient) or scalar value of each of the seven coefficients, as the kernels take them
template <typename T>
void ProxElemDispatch<T>::CoeffArgs(const T* (&ptrs)[7], double (&vals)[7]) const {
  for (int i = 0; i < 7; i++) {
    if (coeffs_[i].size() > 1) {
      if (d_coeffs_[i].size() != coeffs_[i].size()) throw Exception("ProxElemOperation used before Initialize().");
      ptrs[i] = d_coeffs_[i].data(); vals[i] = 0;
    } else { ptrs[i] = nullptr; vals[i] = (double)coeffs_[i][0]; }
  }
}
template <typename T>
void ProxElemDispatch<T>::EvalLocal(T* res, T*, const T* arg, const T*, const T* tau_diag, const T*, T tau, bool invert_tau) {
  const T* ptrs[7]; double vals[7];
  CoeffArgs(ptrs, vals);
  CheckHip(Api<T>::prox_elem(op_, fn_, res, arg, tau_diag, (double)tau, invert_tau ? 1 : 0, this->count_, this->dim_,
                             this->interleaved_ ? 1 : 0, ptrs, vals, CurrentStream()), "prox_elem");
}
/// the device-side argument description of a prox that covers [index, index + size) of its variable (prost_hip_arg_spec)
template <typename T>
static void FillArgSpec(prost_hip_arg_spec& a, const typename Prox<T>::ArgSource& src, size_t index, size_t size) {
  (void)size;
  a.mode = src.mode;
  for (int k = 0; k < 4; k++) a.v[k] = src.v[k] ? src.v[k] + index : nullptr;
  a.s[0] = (double)src.s[0]; a.s[1] = (double)src.s[1];
  a.op = src.op; a.op_rows = src.op_rows; a.op_cols = src.op_cols; a.base = index;
  a.w[0] = src.w[0]; a.w[1] = src.w[1];                        // WHOLE vectors: the operator indexes them
  a.kty_out = src.kty_out ? src.kty_out + index : nullptr;
  a.use[0] = src.use[0]; a.use[1] = src.use[1];
  a.res_ws = src.res_ws; a.res_slot = src.res_slot ? *src.res_slot : 0; a.res_slots_max = src.res_slots_max;
}
/// workgroups (= partial slots) the 16-bytes-per-lane kernel launches for `count` element groups under the cap of the source
template <typename T>
static unsigned ResSlotsOfLaunch(const typename Prox<T>::ArgSource& src, size_t count) {
  const size_t v = 16 / sizeof(T), lanes = count / v;
  size_t g = (lanes + 255) / 256;
  if (g < 1) g = 1;
  if (g > (size_t)1 << 20) g = (size_t)1 << 20;
  return (unsigned)std::min<size_t>(g, src.res_slots_max);
}
template <typename T>
void ProxElemDispatch<T>::EvalMoreauLocal(T* res, const T* arg, const T* tau_diag, T tau, bool invert_tau) {
  const T* ptrs[7]; double vals[7];
  CoeffArgs(ptrs, vals);
  CheckHip(Api<T>::prox_elem_moreau(op_, fn_, res, arg, tau_diag, (double)tau, invert_tau ? 1 : 0, this->count_, this->dim_,
                                    this->interleaved_ ? 1 : 0, ptrs, vals, CurrentStream()), "prox_elem_moreau");
}
template <typename T>
void ProxElemDispatch<T>::EvalSourceLocal(bool moreau, device_vector<T>& result, const typename Prox<T>::ArgSource& src,
                                           const device_vector<T>& tau_diag, T tau, bool invert_tau) {
  const T* ptrs[7]; double vals[7];
  CoeffArgs(ptrs, vals);
  prost_hip_arg_spec a;
  FillArgSpec<T>(a, src, this->index_, this->size_);
  CheckHip(Api<T>::prox_elem_arg(op_, fn_, moreau ? 1 : 0, result.data() + this->index_, &a, tau_diag.data() + this->index_, (double)tau,
                                 invert_tau ? 1 : 0, this->count_, this->dim_, this->interleaved_ ? 1 : 0, ptrs, vals, CurrentStream()), "prox_elem_arg");
  if (src.res_ws && src.res_slot) *src.res_slot += ResSlotsOfLaunch<T>(src, this->count_);
}
template <typename T>
bool ProxElemDispatch<T>::supports_op_source() const {
  const size_t v = 16 / sizeof(T);
  const bool planar = op_ == PROST_OP_1D || !this->interleaved_ || this->dim_ == 1;
  return planar && this->count_ % v == 0 && this->index_ % v == 0;
}
template class ProxElemDispatch<float>;
template class ProxElemDispatch<double>;

// ---- Moreau ----
template <typename T> void ProxMoreau<T>::Initialize() { scaled_arg_.resize(this->size_); conjugate_->Initialize(); }
template <typename T> void ProxMoreau<T>::Release() { conjugate_->Release(); scaled_arg_.clear(); }
static bool g_moreau_fuse = true;
template <typename T> void ProxMoreau<T>::SetFuseElemOperations(bool on) { g_moreau_fuse = on; }
template <typename T>
void ProxMoreau<T>::EvalLocal(T* res, T* res_end, const T* arg, const T* arg_end, const T* tau_diag, const T* tau_end, T tau, bool invert_tau) {
  const size_t n = this->size_;
  if (g_moreau_fuse && res != arg) {
    if (auto* e = dynamic_cast<ProxElemDispatch<T>*>(conjugate_.get())) {        // pre-scale, elem operation and post-scale in one kernel
      e->EvalMoreauLocal(res, arg, tau_diag, tau, invert_tau);
      return;
    }
  }
  CheckHip(Api<T>::moreau_prescale(scaled_arg_.data(), arg, tau_diag, (double)tau, invert_tau ? 1 : 0, n, CurrentStream()), "moreau_prescale");
  conjugate_->EvalLocal(res, res_end, scaled_arg_.data(), scaled_arg_.data() + n, tau_diag, tau_end, tau, !invert_tau);
  CheckHip(Api<T>::moreau_postscale(res, arg, tau_diag, (double)tau, invert_tau ? 1 : 0, n, CurrentStream()), "moreau_postscale");
  (void)arg_end;
}
template <typename T>
bool ProxMoreau<T>::describe(ProxDesc& d) const {
  if (!conjugate_->describe(d) || d.moreau) return false;
  d.moreau = true;
  return true;
}
template <typename T>
bool ProxMoreau<T>::supports_arg_source() const { return g_moreau_fuse && dynamic_cast<ProxElemDispatch<T>*>(conjugate_.get()) != nullptr; }
template <typename T>
void ProxMoreau<T>::EvalFromSource(device_vector<T>& result, const typename Prox<T>::ArgSource& src, const device_vector<T>& tau_diag, T tau, bool invert_tau) {
  auto* e = dynamic_cast<ProxElemDispatch<T>*>(conjugate_.get());
  if (!e) throw Exception("ProxMoreau: only a conjugated elem operation evaluates from an argument source.");
  e->EvalSourceLocal(true, result, src, tau_diag, tau, invert_tau);
}
template class ProxMoreau<float>;
template class ProxMoreau<double>;

// ---- zero ----
template <typename T>
void ProxZero<T>::EvalFromSource(device_vector<T>& result, const typename Prox<T>::ArgSource& src, const device_vector<T>&, T, bool) {
  const size_t i = this->index_;
  void* s = CurrentStream();
  const bool sums = src.res_ws != nullptr && (src.mode == PROST_ARG_PDHG_PRIMAL || src.mode == PROST_ARG_PDHG_DUAL);
  if (src.mode == PROST_ARG_PDHG_PRIMAL && !sums)
    CheckHip(Api<T>::pdhg_primal_arg(result.data() + i, src.v[0] + i, src.v[1] + i, src.v[2] + i, (double)src.s[0], this->size_, s), "primal_arg");
  else if (src.mode == PROST_ARG_PDHG_DUAL && !sums)
    CheckHip(Api<T>::pdhg_dual_arg(result.data() + i, src.v[0] + i, src.v[1] + i, src.v[2] + i, src.v[3] + i, (double)src.s[0], (double)src.s[1], this->size_, s), "dual_arg");
  else if (sums || src.mode == PROST_ARG_PDHG_PRIMAL_OP || src.mode == PROST_ARG_PDHG_DUAL_OP) {
    // the identity as the 1-D operation of the zero function with a = c = 1, b = d = e = 0: ((1 (arg - 0 tau)) / 1 - 0 + 0) / 1 = arg, exactly
    // (elem_operation_1d.hpp:45-58) -- the launch that forms the argument from the operator writes it straight into the result
    prost_hip_arg_spec a;
    FillArgSpec<T>(a, src, i, this->size_);
    const T* ptrs[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const double vals[7] = {1, 0, 1, 0, 0, 0, 0};
    CheckHip(Api<T>::prox_elem_arg(PROST_OP_1D, PROST_FN_ZERO, 0, result.data() + i, &a, src.v[1] + i, (double)src.s[0], 0, this->size_, 1, 0, ptrs, vals, s), "prox_elem_arg");
    if (src.res_ws && src.res_slot) *src.res_slot += ResSlotsOfLaunch<T>(src, this->size_);
  }
  else throw Exception("ProxZero: unknown argument source.");
}
template <typename T>
void ProxZero<T>::EvalLocal(T* res, T*, const T* arg, const T*, const T*, const T*, T, bool) {
  if (res != arg) CheckHip(prost_hip_memcpy_d2d(res, arg, this->size_ * sizeof(T), CurrentStream()), "memcpy_d2d");
}
template class ProxZero<float>;
template class ProxZero<double>;

// ---- epigraph of a quadratic ----
template <typename T>
void ProxIndEpiQuad<T>::Initialize() {
  if (a_.size() != this->count_ && a_.size() != 1) throw Exception("Wrong input: Coefficient a has to have dimension count or 1!");
  for (T& a : a_) if (a <= 0) throw Exception("Wrong input: Coefficient a must be greater 0!");
  if (b_.size() != this->count_ * (this->dim_ - 1)) throw Exception("Wrong input: Coefficient b has to have dimension count*(dim-1)!");
  if (c_.size() != this->count_ && c_.size() != 1) throw Exception("Wrong input: Coefficient c has to have dimension count or 1!");
  d_a_ = a_; d_b_ = b_; d_c_ = c_;
}
template <typename T> void ProxIndEpiQuad<T>::Release() { d_a_.clear(); d_b_.clear(); d_c_.clear(); }
template <typename T>
void ProxIndEpiQuad<T>::EvalLocal(T* res, T*, const T* arg, const T*, const T*, const T*, T, bool) {
  if (d_b_.size() != b_.size()) throw Exception("ProxIndEpiQuad used before Initialize().");
  CheckHip(Api<T>::prox_epi_quad(res, arg, this->count_, this->dim_, a_.size() != 1 ? d_a_.data() : nullptr, (double)a_[0], d_b_.data(),
                                 c_.size() != 1 ? d_c_.data() : nullptr, (double)c_[0], CurrentStream()), "prox_epi_quad");
}
template class ProxIndEpiQuad<float>;
template class ProxIndEpiQuad<double>;

// ---- elem_operation:ind_sum ----
template <typename T>
void ProxElemIndSum<T>::EvalLocal(T* res, T*, const T* arg, const T*, const T*, const T*, T, bool) {
  CheckHip(Api<T>::prox_elem_ind_sum(res, arg, this->count_, this->dim_, this->interleaved_ ? 1 : 0, CurrentStream()), "prox_elem_ind_sum");
}
template class ProxElemIndSum<float>;
template class ProxElemIndSum<double>;

// ---- elem_operation:ind_simplex ----
template <typename T>
void ProxElemIndSimplex<T>::EvalLocal(T* res, T*, const T* arg, const T*, const T*, const T*, T, bool) {
  if (work_.size() != this->count_ * this->dim_) throw Exception("ProxElemIndSimplex used before Initialize().");
  CheckHip(Api<T>::prox_elem_ind_simplex(res, arg, work_.data(), this->count_, this->dim_, this->interleaved_ ? 1 : 0, CurrentStream()), "prox_elem_ind_simplex");
}
template class ProxElemIndSimplex<float>;
template class ProxElemIndSimplex<double>;

// ---- transform ----
template <typename T>
void ProxTransform<T>::Initialize() {
  for (T& a : host_[0])
    if (a == 0) throw Exception("ProxTransform: Vector 'a' isn't allowed to contain zero element. (Division by zero)");
  for (int k = 0; k < 5; k++) {
    if (host_[k].empty()) throw Exception("Empty vector passed.");
    if (host_[k].size() > 1) {
      if (host_[k].size() < this->size_) throw Exception("Size of coefficients should be either 1 or count.");
      dev_[k] = host_[k];
    }
  }
  scaled_arg_.resize(this->size_);
  scaled_tau_.resize(this->size_);
  inner_fn_->Initialize();
}
template <typename T>
void ProxTransform<T>::Release() {
  inner_fn_->Release();
  for (int k = 0; k < 5; k++) dev_[k].clear();
  scaled_arg_.clear(); scaled_tau_.clear();
}
template <typename T>
size_t ProxTransform<T>::gpu_mem_amount() const {
  size_t mem = inner_fn_->gpu_mem_amount() + 2 * this->size_ * sizeof(T);
  for (int k = 0; k < 5; k++) if (host_[k].size() > 1) mem += host_[k].size() * sizeof(T);
  return mem;
}
template <typename T>
void ProxTransform<T>::EvalLocal(T* res, T* res_end, const T* arg, const T*, const T* tau_diag, const T*, T tau, bool invert_tau) {
  const size_t n = this->size_;
  if (scaled_arg_.size() != n) throw Exception("ProxTransform used before Initialize().");
  const T* ptrs[5]; double vals[5];
  for (int k = 0; k < 5; k++) {
    ptrs[k] = host_[k].size() > 1 ? dev_[k].data() : nullptr;
    vals[k] = (double)host_[k][0];
  }
  void* s = CurrentStream();
  CheckHip(Api<T>::transform_prescale(scaled_arg_.data(), scaled_tau_.data(), arg, tau_diag, ptrs, vals, (double)tau, invert_tau ? 1 : 0, n, s), "transform_prescale");
  // the inner prox sees local ranges: its own index is not applied (prox_transform.cu:202-210)
  inner_fn_->EvalLocal(res, res_end, scaled_arg_.data(), scaled_arg_.data() + n, scaled_tau_.data(), scaled_tau_.data() + n, (T)1, false);
  CheckHip(Api<T>::transform_postscale(res, ptrs[0], vals[0], ptrs[1], vals[1], n, s), "transform_postscale");
}
template class ProxTransform<float>;
template class ProxTransform<double>;

// ---- permute ----
template <typename T>
void ProxPermute<T>::Initialize() {
  permuted_arg_.resize(this->size_);
  perm_ = perm_host_;
  if (perm_host_.size() != base_prox_->size()) {
    std::stringstream ss;
    ss << "Permutation vector has wrong size (" << perm_host_.size() << ") instead of " << base_prox_->size() << ".";
    throw Exception(ss.str());
  }
  for (int32_t v : perm_host_)      // the reference does not check; an out-of-range entry would read / write out of bounds
    if (v < 0 || (size_t)v >= perm_host_.size()) throw Exception("Permutation vector has an entry outside [0, size).");
  base_prox_->Initialize();
}
template <typename T> void ProxPermute<T>::Release() { base_prox_->Release(); perm_.clear(); permuted_arg_.clear(); }
template <typename T>
void ProxPermute<T>::EvalLocal(T* res, T* res_end, const T* arg, const T*, const T* tau_diag, const T* tau_end, T tau, bool invert_tau) {
  const size_t n = perm_host_.size();
  if (permuted_arg_.size() != this->size_) throw Exception("ProxPermute used before Initialize().");
  void* s = CurrentStream();
  // permuted argument into `res`, prox of it into permuted_arg_, scattered back into `res`; tau_diag is
  // NOT permuted (prox_permute.cu:113-143)
  CheckHip(Api<T>::permute(res, arg, perm_.data(), n, 0, s), "permute");
  base_prox_->EvalLocal(permuted_arg_.data(), permuted_arg_.data() + n, res, res_end, tau_diag, tau_end, tau, invert_tau);
  CheckHip(Api<T>::permute(res, permuted_arg_.data(), perm_.data(), n, 1, s), "permute");
}
template class ProxPermute<float>;
template class ProxPermute<double>;

// ---- halfspace ----
template <typename T>
void ProxIndHalfspace<T>::Initialize() {
  if (a_.size() != this->count_ * this->dim_ && a_.size() != this->dim_) throw Exception("Wrong input: Coefficient a has to have dimension count*dim or dim!");
  if (b_.size() != this->count_ && b_.size() != 1) throw Exception("Wrong input: Coefficient b has to have dimension count or 1!");
  d_a_ = a_; d_b_ = b_;
}
template <typename T>
void ProxIndHalfspace<T>::EvalLocal(T* res, T*, const T* arg, const T*, const T*, const T*, T, bool) {
  if (d_a_.size() != a_.size()) throw Exception("ProxIndHalfspace used before Initialize().");
  CheckHip(Api<T>::prox_ind_halfspace(res, arg, this->count_, this->dim_, d_a_.data(), a_.size(), d_b_.data(), b_.size(), CurrentStream()), "prox_ind_halfspace");
}
template class ProxIndHalfspace<float>;
template class ProxIndHalfspace<double>;

// ---- second-order cone ----
template <typename T>
void ProxIndSOC<T>::Initialize() {
  if (alpha_ != 1) throw Exception("ProxIndSOC: Only alpha = 1 implemented right now.");
  if (this->dim_ < 1) throw Exception("ProxIndSOC: dim must be at least 1.");
}
template <typename T>
void ProxIndSOC<T>::EvalLocal(T* res, T*, const T* arg, const T*, const T*, const T*, T, bool) {
  CheckHip(Api<T>::prox_ind_soc(res, arg, this->count_, this->dim_, CurrentStream()), "prox_ind_soc");
}
template class ProxIndSOC<float>;
template class ProxIndSOC<double>;

// ---- index-family sum constraints ----
template <typename T>
void ProxIndSum<T>::Initialize() {
  if (count_ * dim_ != inds_.size()) throw Exception("ProxIndSum: dimensions dont fit");
  if (two_ && count_2_ * dim_2_ != inds_2_.size()) throw Exception("ProxIndSum: dimensions dont fit");
  for (uint64_t v : inds_) if (v >= this->size_) throw Exception("ProxIndSum: index outside the prox range.");   // unchecked in the reference
  for (uint64_t v : inds_2_) if (v >= this->size_) throw Exception("ProxIndSum: index outside the prox range.");
  d_inds_ = std::vector<int64_t>(inds_.begin(), inds_.end());
  if (two_) d_inds_2_ = std::vector<int64_t>(inds_2_.begin(), inds_2_.end());
}
template <typename T>
void ProxIndSum<T>::EvalLocal(T* res, T*, const T* arg, const T*, const T* tau_diag, const T*, T tau, bool invert_tau) {
  if (d_inds_.size() != inds_.size()) throw Exception("ProxIndSum used before Initialize().");
  void* s = CurrentStream();
  if (res != arg) CheckHip(prost_hip_memcpy_d2d(res, arg, this->size_ * sizeof(T), s), "memcpy_d2d");       // "zero prox on other indices" :119
  CheckHip(Api<T>::prox_ind_sum(res, arg, tau_diag, reinterpret_cast<const uint64_t*>(d_inds_.data()), count_, dim_, (double)sum_, (double)tau,
                                invert_tau ? 1 : 0, s), "prox_ind_sum");
  if (two_)
    CheckHip(Api<T>::prox_ind_sum(res, arg, tau_diag, reinterpret_cast<const uint64_t*>(d_inds_2_.data()), count_2_, dim_2_, (double)sum_2_,
                                  (double)tau, invert_tau ? 1 : 0, s), "prox_ind_sum");
}
template class ProxIndSum<float>;
template class ProxIndSum<double>;

}  // namespace prost
