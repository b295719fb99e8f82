// prost/prox/prox_elem_operation.hpp -- a prost::Prox from a user-written elementwise operation.
//
// Plugin contract of the reference's include/prost/prox/prox_elem_operation.hpp:32-110:
//   ProxElemOperation<T, ELEM_OPERATION>(index, count, dim, interleaved, diagsteps)                 kCoeffsCount == 0
//   ProxElemOperation<T, ELEM_OPERATION>(index, count, dim, interleaved, diagsteps, coeffs)         kCoeffsCount != 0
// with coeffs = std::array<std::vector<T>, kCoeffsCount>, every entry one value (shared by all element groups) or
// `count` values (one per group; uploaded by Initialize()).  ELEM_OPERATION::kDim > 0 overrides `dim`.
//
// The member definitions and the gfx950 kernel live in prox_elem_operation.inl, which the plugin's .hip source
// includes (hipcc; the reference's prox_elem_operation.cu:27 does the same before its explicit instantiations).
// Kernels are enqueued on prost::CurrentStream(); there is no device synchronisation after a prox (the reference
// synchronises after every launch, prox_elem_operation.inl:128, :187).
//
// (The 28 built-in operations -- elem_operation:1d:* / elem_operation:norm2:* -- are not instances of this template:
// they run as ProxElemDispatch<T>, one kernel with wave-uniform run-time function ids that the PDHG backend can fuse
// with its argument passes; proxes.hpp.  The same operations are available to plugin authors as templates in
// prost/prox/elemop/elem_operation_1d.hpp, elem_operation_norm2.hpp and function_1d.hpp.)
#ifndef PROST_PROX_ELEM_OPERATION_HPP_
#define PROST_PROX_ELEM_OPERATION_HPP_
#include <array>
#include <type_traits>
#include <vector>

#include "prost/device_vector.hpp"
#include "prost/prox/prox_separable_sum.hpp"

namespace prost {

template <typename T, class ELEM_OPERATION>
struct ElemOpCoefficients;

template <typename T, class ELEM_OPERATION, class ENABLE = void>
class ProxElemOperation {};

/// operations without coefficients (prox_elem_operation.hpp:35-62)
template <typename T, class ELEM_OPERATION>
class ProxElemOperation<T, ELEM_OPERATION, typename std::enable_if<ELEM_OPERATION::kCoeffsCount == 0>::type> : public ProxSeparableSum<T> {
 public:
  ProxElemOperation(size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps)
      : ProxSeparableSum<T>(index, count, (ELEM_OPERATION::kDim <= 0) ? dim : ELEM_OPERATION::kDim, interleaved, diagsteps) {}
  virtual size_t gpu_mem_amount() const { return 0; }
  /// MI355X addition: the kernel reads the step size from device memory when asked to (prox.hpp: StepView) -- problems whose proxes are
  /// user-written operations run goldstein / boyd without a host wait per iteration as well
  virtual bool takes_step_view() const { return true; }

 protected:
  virtual void EvalLocal(T* result_beg, T* result_end, const T* arg_beg, const T* arg_end, const T* tau_beg, const T* tau_end,
                         T tau, bool invert_tau);
  virtual void EvalLocalStepView(T* result_beg, T* result_end, const T* arg_beg, const T* arg_end, const T* tau_beg, const T* tau_end,
                                 const typename Prox<T>::StepView& view, bool invert_tau);
};

/// operations with kCoeffsCount scalar-or-per-group coefficients (prox_elem_operation.hpp:64-110)
template <typename T, class ELEM_OPERATION>
class ProxElemOperation<T, ELEM_OPERATION, typename std::enable_if<ELEM_OPERATION::kCoeffsCount != 0>::type> : public ProxSeparableSum<T> {
 public:
  ProxElemOperation(size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps,
                    std::array<std::vector<T>, ELEM_OPERATION::kCoeffsCount> coeffs)
      : ProxSeparableSum<T>(index, count, (ELEM_OPERATION::kDim <= 0) ? dim : ELEM_OPERATION::kDim, interleaved, diagsteps), coeffs_(coeffs) {}

  /// uploads the per-group coefficient vectors (prox_elem_operation.inl:200-222)
  virtual void Initialize();
  virtual void Release();
  virtual size_t gpu_mem_amount() const {
    size_t mem = 0;
    for (size_t i = 0; i < ELEM_OPERATION::kCoeffsCount; i++)
      if (coeffs_[i].size() > 1) mem += this->count_ * sizeof(T);
    return mem;
  }
  virtual bool takes_step_view() const { return true; }

 protected:
  virtual void EvalLocal(T* result_beg, T* result_end, const T* arg_beg, const T* arg_end, const T* tau_beg, const T* tau_end,
                         T tau, bool invert_tau);
  virtual void EvalLocalStepView(T* result_beg, T* result_end, const T* arg_beg, const T* arg_end, const T* tau_beg, const T* tau_end,
                                 const typename Prox<T>::StepView& view, bool invert_tau);

 private:
  void KernelCoefficients(ElemOpCoefficients<T, ELEM_OPERATION>& coeffs) const;
  std::array<std::vector<T>, ELEM_OPERATION::kCoeffsCount> coeffs_;
  std::array<device_vector<T>, ELEM_OPERATION::kCoeffsCount> d_coeffs_;
};

/// kernel argument: per coefficient a device vector (one value per element group) or, where dev_p is null, the value
/// shared by all groups (prox_elem_operation.hpp:112-117)
template <typename T, class ELEM_OPERATION>
struct ElemOpCoefficients {
  static const size_t kSlots = ELEM_OPERATION::kCoeffsCount ? ELEM_OPERATION::kCoeffsCount : 1;
  const T* dev_p[kSlots];
  T val[kSlots];
};

}  // namespace prost
#endif
