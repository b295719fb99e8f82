// OUT-OF-TREE block and prox written with the REFERENCE's signatures -- thrust::device_vector<T>::iterator ranges
// (include/prost/linop/block.hpp:66-77, include/prost/prox/prox.hpp:117-126) and thrust algorithms in the bodies, the way
// src/linop/block_zero.cu / src/prox/prox_zero.cu and user code written after them look -- compiled against
// include/prost/compat/thrust_ranges.hpp + rocThrust:
//   test:thrust:scaled_identity   K = s I:   EvalLocalAdd:  res += s rhs   (thrust::transform over the ranges)
//   test:thrust:soft_threshold    prox of lambda |x|_1 with per-element steps (a zip over arg and tau_diag)
// tests/test_plugins.py evaluates both against the oracle running block.identity(s) and sum_1d('abs', 1, 0, lambda).
#include <thrust/functional.h>
#include <thrust/iterator/zip_iterator.h>
#include <thrust/transform.h>
#include <thrust/tuple.h>

#include "prost/compat/thrust_ranges.hpp"
#include "prost/factory.hpp"

namespace {

template <typename T>
struct axpy_functor {
  T s;
  __host__ __device__ T operator()(const T& res, const T& rhs) const { return res + s * rhs; }
};

// written against the reference's Block<T> (block.hpp:37-83): iterator ranges in, thrust::transform inside
template <typename T>
class ThrustScaledIdentity : public prost::compat::Block<T> {
 public:
  ThrustScaledIdentity(size_t row, size_t col, size_t n, T s) : prost::compat::Block<T>(row, col, n, n), s_(s) {}
  virtual T row_sum(size_t, T alpha) const { return std::pow(std::abs(s_), alpha); }
  virtual T col_sum(size_t, T alpha) const { return std::pow(std::abs(s_), alpha); }
  virtual size_t gpu_mem_amount() const { return 0; }

 protected:
  virtual void EvalLocalAdd(const typename thrust::device_vector<T>::iterator& res_begin, const typename thrust::device_vector<T>::iterator& res_end,
                            const typename thrust::device_vector<T>::const_iterator& rhs_begin, const typename thrust::device_vector<T>::const_iterator& rhs_end) {
    thrust::transform(res_begin, res_end, rhs_begin, res_begin, axpy_functor<T>{s_});
  }
  virtual void EvalAdjointLocalAdd(const typename thrust::device_vector<T>::iterator& res_begin, const typename thrust::device_vector<T>::iterator& res_end,
                                   const typename thrust::device_vector<T>::const_iterator& rhs_begin, const typename thrust::device_vector<T>::const_iterator& rhs_end) {
    thrust::transform(res_begin, res_end, rhs_begin, res_begin, axpy_functor<T>{s_});
  }

 private:
  T s_;
};

// step = lambda tau' with tau' = tau tau_diag (or its reciprocal formed in double, elem_operation_1d.hpp:38-41); Function1DAbs (function_1d.hpp:47-60)
template <typename T>
struct soft_threshold_functor {
  T tau, lambda;
  bool invert_tau;
  __host__ __device__ T operator()(const thrust::tuple<T, T>& arg_tau) const {
    const T x = thrust::get<0>(arg_tau), td = thrust::get<1>(arg_tau);
    const T t = invert_tau ? (T)(1. / (double)(tau * td)) : (tau * td);
    const T step = lambda * t;
    return x >= step ? x - step : (x <= -step ? x + step : (T)0);
  }
};

// written against the reference's Prox<T> (prox.hpp:39-135)
template <typename T>
class ThrustSoftThreshold : public prost::compat::Prox<T> {
 public:
  ThrustSoftThreshold(size_t index, size_t size, bool diagsteps, T lambda) : prost::compat::Prox<T>(index, size, diagsteps), lambda_(lambda) {}
  virtual size_t gpu_mem_amount() const { return 0; }

 protected:
  virtual void EvalLocal(const typename thrust::device_vector<T>::iterator& result_beg, const typename thrust::device_vector<T>::iterator& result_end,
                         const typename thrust::device_vector<T>::const_iterator& arg_beg, const typename thrust::device_vector<T>::const_iterator& arg_end,
                         const typename thrust::device_vector<T>::const_iterator& tau_beg, const typename thrust::device_vector<T>::const_iterator& tau_end,
                         T tau, bool invert_tau) {
    thrust::transform(thrust::make_zip_iterator(thrust::make_tuple(arg_beg, tau_beg)), thrust::make_zip_iterator(thrust::make_tuple(arg_end, tau_end)),
                      result_beg, soft_threshold_functor<T>{tau, lambda_, invert_tau});
  }

 private:
  T lambda_;
};

// block cell {name, row, col, data}, data = {n, s}; prox cell {name, idx, size, diagsteps, data}, data = {lambda}   (custom.cpp:11-28)
template <typename T>
prost::Block<T>* CreateScaledIdentity(size_t row, size_t col, const prost_value* data) {
  return new ThrustScaledIdentity<T>(row, col, (size_t)prost::GetScalarFromCell(data, 0), (T)prost::GetScalarFromCell(data, 1));
}
template <typename T>
prost::Prox<T>* CreateSoftThreshold(size_t idx, size_t size, bool diagsteps, const prost_value* data) {
  return new ThrustSoftThreshold<T>(idx, size, diagsteps, (T)prost::GetScalarFromCell(data, 0));
}

const bool registered = [] {
  prost::Factory<float>::block_reg()["test:thrust:scaled_identity"] = CreateScaledIdentity<float>;
  prost::Factory<double>::block_reg()["test:thrust:scaled_identity"] = CreateScaledIdentity<double>;
  prost::Factory<float>::prox_reg()["test:thrust:soft_threshold"] = CreateSoftThreshold<float>;
  prost::Factory<double>::prox_reg()["test:thrust:soft_threshold"] = CreateSoftThreshold<double>;
  return true;
}();

}  // namespace
