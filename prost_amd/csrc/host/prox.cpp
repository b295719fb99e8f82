// prox.cpp -- proximal operators of the prost host library (calls prost_hip.h only).
#include <chrono>

#include "hipapi.hpp"
#include "prost/prox/proxes.hpp"

namespace prost {

template <typename T>
void Prox<T>::Eval(device_vector<T>& result, const device_vector<T>& arg, const device_vector<T>& tau_diag, T tau, bool invert_tau) {
  EvalLocal(result.data() + index_, result.data() + index_ + size_, arg.data() + index_, arg.data() + index_ + size_,
            tau_diag.data() + index_, tau_diag.data() + index_ + size_, tau, invert_tau);
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
ProxElemOperation<T>::ProxElemOperation(int op, int fn, size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps,
                                        const std::array<std::vector<T>, 7>& coeffs)
    : ProxSeparableSum<T>(index, count, op == PROST_OP_1D ? 1 : dim, interleaved, diagsteps), op_(op), fn_(fn), coeffs_(coeffs) {
  for (int i = 0; i < 7; i++)
    if (coeffs_[i].empty()) throw Exception("Empty vector passed.");
}
template <typename T>
void ProxElemOperation<T>::Initialize() {
  for (int i = 0; i < 7; i++)
    if (coeffs_[i].size() > 1) {
      if (coeffs_[i].size() < this->count_) throw Exception("Size of coefficients should be either 1 or count.");
      d_coeffs_[i] = coeffs_[i];
    }
}
template <typename T> void ProxElemOperation<T>::Release() { for (int i = 0; i < 7; i++) d_coeffs_[i].clear(); }
template <typename T>
size_t ProxElemOperation<T>::gpu_mem_amount() const {
  size_t mem = 0;
  for (int i = 0; i < 7; i++) if (coeffs_[i].size() > 1) mem += this->count_ * sizeof(T);
  return mem;
}
template <typename T>
bool ProxElemOperation<T>::describe(ProxDesc& d) const {
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
template <typename T>
void ProxElemOperation<T>::EvalLocal(T* res, T*, const T* arg, const T*, const T* tau_diag, const T*, T tau, bool invert_tau) {
  const T* ptrs[7]; double vals[7];
  for (int i = 0; i < 7; i++) {
    if (coeffs_[i].size() > 1) {
      if (d_coeffs_[i].size() != coeffs_[i].size()) throw Exception("ProxElemOperation used before Initialize().");
      ptrs[i] = d_coeffs_[i].data(); vals[i] = 0;
    } else { ptrs[i] = nullptr; vals[i] = (double)coeffs_[i][0]; }
  }
  CheckHip(Api<T>::prox_elem(op_, fn_, res, arg, tau_diag, (double)tau, invert_tau ? 1 : 0, this->count_, this->dim_,
                             this->interleaved_ ? 1 : 0, ptrs, vals, CurrentStream()), "prox_elem");
}
template class ProxElemOperation<float>;
template class ProxElemOperation<double>;

// ---- Moreau ----
template <typename T> void ProxMoreau<T>::Initialize() { scaled_arg_.resize(this->size_); conjugate_->Initialize(); }
template <typename T> void ProxMoreau<T>::Release() { conjugate_->Release(); scaled_arg_.clear(); }
template <typename T>
void ProxMoreau<T>::EvalLocal(T* res, T* res_end, const T* arg, const T* arg_end, const T* tau_diag, const T* tau_end, T tau, bool invert_tau) {
  const size_t n = this->size_;
  CheckHip(Api<T>::moreau_prescale(scaled_arg_.data(), arg, tau_diag, (double)tau, invert_tau ? 1 : 0, n, CurrentStream()), "moreau_prescale");
  conjugate_->EvalLocal(res, res_end, scaled_arg_.data(), scaled_arg_.data() + n, tau_diag, tau_end, tau, !invert_tau);
  CheckHip(Api<T>::moreau_postscale(res, arg, tau_diag, (double)tau, invert_tau ? 1 : 0, n, CurrentStream()), "moreau_postscale");
  (void)arg_end;
}
template class ProxMoreau<float>;
template class ProxMoreau<double>;

// ---- zero ----
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

}  // namespace prost
