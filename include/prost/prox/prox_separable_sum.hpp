// prost/prox/prox_separable_sum.hpp -- sum of `count` dim-dimensional functions
// (reference prox_separable_sum.hpp:47-86; interleaved: x0 y0 x1 y1 ..., else x0 x1 .. y0 y1 ..).
#ifndef PROST_PROX_SEPARABLE_SUM_HPP_
#define PROST_PROX_SEPARABLE_SUM_HPP_
#include "prost/prox/prox.hpp"

namespace prost {

template <typename T>
class ProxSeparableSum : public Prox<T> {
 public:
  ProxSeparableSum(size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps)
      : Prox<T>(index, count * dim, diagsteps), count_(count), dim_(dim), interleaved_(interleaved) {}
  size_t dim() const { return dim_; }
  size_t count() const { return count_; }
  bool interleaved() const { return interleaved_; }
  virtual void get_separable_structure(std::vector<std::tuple<size_t, size_t, size_t>>& sep) {
    if (interleaved_) for (size_t i = 0; i < count_; i++) sep.push_back(std::tuple<size_t, size_t, size_t>(this->index_ + i * dim_, dim_, 1));
    else for (size_t i = 0; i < count_; i++) sep.push_back(std::tuple<size_t, size_t, size_t>(this->index_ + i, dim_, count_));
  }
  virtual bool average_uniform(T& value) const {           // every group: dim_ copies of `value`, summed in order, divided
    T avg = 0;
    for (size_t c = 0; c < dim_; c++) avg += value;
    avg /= static_cast<T>(dim_);
    value = avg;
    return true;
  }
  virtual void average_preconditioner(std::vector<T>& precond) {
    T* base = precond.data() + this->index_;
    if (interleaved_) {
      for (size_t i = 0; i < count_; i++) {
        T avg = 0;
        for (size_t c = 0; c < dim_; c++) avg += base[i * dim_ + c];
        avg /= static_cast<T>(dim_);
        for (size_t c = 0; c < dim_; c++) base[i * dim_ + c] = avg;
      }
    } else {                                    // planar: component c of element i at i + c * count
      std::vector<T> avg(count_);
      const T cnt = static_cast<T>(dim_);
      ParallelFor(count_, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) avg[i] = 0;
        for (size_t c = 0; c < dim_; c++) { const T* src = base + c * count_; for (size_t i = b; i < e; i++) avg[i] += src[i]; }
        for (size_t i = b; i < e; i++) avg[i] /= cnt;
        for (size_t c = 0; c < dim_; c++) { T* dst = base + c * count_; for (size_t i = b; i < e; i++) dst[i] = avg[i]; }
      });
    }
  }

 protected:
  size_t count_, dim_;
  bool interleaved_;
};

}  // namespace prost
#endif
