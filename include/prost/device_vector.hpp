// prost/device_vector.hpp -- owning 1-D HBM buffer used where the reference uses
// thrust::device_vector<T>.  Allocation / copies go through the C ABI of prost_hip.h only.
#ifndef PROST_DEVICE_VECTOR_HPP_
#define PROST_DEVICE_VECTOR_HPP_
#include <cstddef>
#include <utility>
#include <vector>

namespace prost {

template <typename T>
class device_vector {
 public:
  device_vector() : data_(nullptr), size_(0) {}
  explicit device_vector(size_t n) : data_(nullptr), size_(0) { resize(n); }
  device_vector(size_t n, T fill) : data_(nullptr), size_(0) { resize(n, fill); }
  device_vector(const device_vector&) = delete;
  device_vector& operator=(const device_vector&) = delete;
  device_vector(device_vector&& o) noexcept : data_(o.data_), size_(o.size_) { o.data_ = nullptr; o.size_ = 0; }
  device_vector& operator=(device_vector&& o) noexcept { swap(o); return *this; }
  ~device_vector();

  /// (re)allocates and zero-fills (thrust resize(n, 0) semantics of backend_pdhg.cu:210-221)
  void resize(size_t n);
  void resize(size_t n, T fill);
  void clear();
  void swap(device_vector& o) { std::swap(data_, o.data_); std::swap(size_, o.size_); }

  device_vector& operator=(const std::vector<T>& host);   // upload (allocates)
  void copy_to(std::vector<T>& host) const;               // download, synchronous
  void copy_from(const device_vector& o);                 // device to device, async

  T* data() { return data_; }
  const T* data() const { return data_; }
  T* begin() { return data_; }
  const T* begin() const { return data_; }
  T* end() { return data_ + size_; }
  const T* end() const { return data_ + size_; }
  size_t size() const { return size_; }
  bool empty() const { return size_ == 0; }

 private:
  T* data_;
  size_t size_;
};

}  // namespace prost
#endif
