// An OUT-OF-TREE linear-operator block: K = scale * I (n x n).  Compiled by plain g++ against include/ and
// libprost.so / libprost_hip.so, never part of the library -- the mechanism of the reference's custom.cpp:11-28 +
// cmake/CustomSources.cmake.example:1-26 (user sources registering themselves in the factory registries).
// The same operator inside the library is block.diags(n, n, [scale], [0]) (block_diags.cu:36-162), which is what the
// tests compare it with.
#include <cmath>

#include "prost/factory.hpp"
#include "prost_hip.h"

namespace {

template <typename T> struct Axpy;
template <> struct Axpy<float> { static int run(float* y, const float* x, double a, size_t n, void* s) { return prost_hip_axpy_f32(y, x, a, n, s); } };
template <> struct Axpy<double> { static int run(double* y, const double* x, double a, size_t n, void* s) { return prost_hip_axpy_f64(y, x, a, n, s); } };

template <typename T>
class ScaledIdentityBlock : public prost::Block<T> {
 public:
  ScaledIdentityBlock(size_t row, size_t col, size_t n, double scale) : prost::Block<T>(row, col, n, n), scale_(scale) {}
  // sum_j |K_ij|^alpha of a local row / column (block.hpp:71-75): one entry per row and column
  T row_sum(size_t, T alpha) const override { return std::pow(std::abs(static_cast<T>(scale_)), alpha); }
  T col_sum(size_t, T alpha) const override { return std::pow(std::abs(static_cast<T>(scale_)), alpha); }
  size_t gpu_mem_amount() const override { return 0; }

 protected:
  // accumulate semantics (block.hpp:59-69); ranges are HBM pointers; work is enqueued on prost::CurrentStream()
  void EvalLocalAdd(T* res_begin, T* res_end, const T* rhs_begin, const T*) override {
    prost::CheckHip(Axpy<T>::run(res_begin, rhs_begin, scale_, (size_t)(res_end - res_begin), prost::CurrentStream()), "axpy");
  }
  void EvalAdjointLocalAdd(T* res_begin, T* res_end, const T* rhs_begin, const T*) override {
    prost::CheckHip(Axpy<T>::run(res_begin, rhs_begin, scale_, (size_t)(res_end - res_begin), prost::CurrentStream()), "axpy");
  }

 private:
  double scale_;
};

// block cell {name, row, col, data}, data = {n, scale}   (custom.cpp:19-27: name -> factory(row, col, data))
template <typename T>
prost::Block<T>* CreateScaledIdentity(size_t row, size_t col, const prost_value* data) {
  const size_t n = (size_t)prost::GetScalarFromCell(data, 0);
  const double scale = prost::GetScalarFromCell(data, 1);
  if (n == 0) throw prost::Exception("scaled_identity: n must be positive.");
  return new ScaledIdentityBlock<T>(row, col, n, scale);
}

const bool registered = [] {
  prost::Factory<float>::block_reg()["test:scaled_identity"] = CreateScaledIdentity<float>;
  prost::Factory<double>::block_reg()["test:scaled_identity"] = CreateScaledIdentity<double>;
  return true;
}();

}  // namespace
