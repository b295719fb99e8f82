// An OUT-OF-TREE proximal operator with its OWN gfx950 kernel: prox of lambda * |x|_1 (soft thresholding), a
// prost::ProxSeparableSum subclass (prox_separable_sum.hpp:47-86) registered in the prox registry as custom.cpp:11-17
// does.  Inside the library the same function is sum_1d('abs', 1, 0, lambda) (elem_operation_1d.hpp:36-59 +
// function_1d.hpp:47-60), which the tests compare it with.
#include <hip/hip_runtime.h>

#include "prost/factory.hpp"
#include "prost/prox/prox_separable_sum.hpp"

namespace {

// step = lambda * tau' with tau' = tau * tau_diag (or its reciprocal formed in double, elem_operation_1d.hpp:38-41);
// res = x - step | x + step | 0   (Function1DAbs)
template <typename T>
__global__ void __launch_bounds__(256) soft_threshold_kernel(T* __restrict__ res, const T* __restrict__ arg, const T* __restrict__ tau_diag,
                                                             T tau, T lambda, bool invert_tau, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T t = invert_tau ? (T)(1. / (double)(tau * tau_diag[i])) : (tau * tau_diag[i]);
  const T step = lambda * t;
  const T x = arg[i];
  res[i] = x >= step ? x - step : (x <= -step ? x + step : (T)0);
}

template <typename T>
class SoftThresholdProx : public prost::ProxSeparableSum<T> {
 public:
  SoftThresholdProx(size_t index, size_t count, bool diagsteps, double lambda)
      : prost::ProxSeparableSum<T>(index, count, 1, false, diagsteps), lambda_(lambda) {}
  size_t gpu_mem_amount() const override { return 0; }

 protected:
  void EvalLocal(T* result_beg, T* result_end, const T* arg_beg, const T*, const T* tau_beg, const T*, T tau, bool invert_tau) override {
    const size_t n = (size_t)(result_end - result_beg);
    if (n == 0) return;
    hipLaunchKernelGGL((soft_threshold_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)prost::CurrentStream(),
                       result_beg, arg_beg, tau_beg, tau, (T)lambda_, invert_tau, n);
    if (hipGetLastError() != hipSuccess) throw prost::Exception("soft_threshold: kernel launch failed.");
  }

 private:
  double lambda_;
};

// prox cell {name, idx, size, diagsteps, data}, data = {lambda}   (custom.cpp:11-17: name -> factory(idx, size, diagsteps, data))
template <typename T>
prost::Prox<T>* CreateSoftThreshold(size_t idx, size_t size, bool diagsteps, const prost_value* data) {
  const double lambda = prost::GetScalarFromCell(data, 0);
  if (lambda < 0) throw prost::Exception("soft_threshold: lambda must not be negative.");
  return new SoftThresholdProx<T>(idx, size, diagsteps, lambda);
}

const bool registered = [] {
  prost::Factory<float>::prox_reg()["test:soft_threshold"] = CreateSoftThreshold<float>;
  prost::Factory<double>::prox_reg()["test:soft_threshold"] = CreateSoftThreshold<double>;
  return true;
}();

}  // namespace
