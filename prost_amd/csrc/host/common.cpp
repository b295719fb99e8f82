// common.cpp -- device_vector, stream handling and small host helpers of the prost host library.
#include <algorithm>
#include <chrono>
#include <exception>
#include <map>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "hipapi.hpp"
#include "prost/common.hpp"
#include "prost/device_vector.hpp"

namespace prost {

static thread_local void* g_stream = nullptr;
void* CurrentStream() { return g_stream; }
void SetCurrentStream(void* s) { g_stream = s; }

void CheckHip(int rc, const char* what) {
  if (rc != 0) throw Exception(std::string(what) + " failed: " + prost_hip_last_error());
}

std::string get_version() { return "prost-mi355x 0.1 (gfx950)"; }

// ---- device_vector ----
template <typename T> device_vector<T>::~device_vector() { if (data_) prost_hip_free(data_); }
template <typename T> void device_vector<T>::clear() { if (data_) prost_hip_free(data_); data_ = nullptr; size_ = 0; }
template <typename T>
void device_vector<T>::resize(size_t n) {
  if (n != size_) {
    clear();
    if (n) {
      void* p = nullptr;
      int rc = prost_hip_malloc(&p, n * sizeof(T));
      if (rc != 0) throw Exception(std::string("Out of memory: ") + prost_hip_last_error());
      data_ = static_cast<T*>(p);
      size_ = n;
    }
  }
  if (n) CheckHip(prost_hip_memset(data_, 0, n * sizeof(T), CurrentStream()), "memset");
}
template <typename T>
void device_vector<T>::resize(size_t n, T fill) {
  resize(n);
  if (n && fill != T(0)) { std::vector<T> h(n, fill); *this = h; }
}
template <typename T>
device_vector<T>& device_vector<T>::operator=(const std::vector<T>& host) {
  if (host.size() != size_) { clear(); if (!host.empty()) { void* p = nullptr; int rc = prost_hip_malloc(&p, host.size() * sizeof(T)); if (rc != 0) throw Exception(std::string("Out of memory: ") + prost_hip_last_error()); data_ = static_cast<T*>(p); size_ = host.size(); } }
  if (size_ * sizeof(T) >= ((size_t)4 << 20)) {
    // through the pinned staging buffers: host threads copy piece k + 1 while piece k is on the bus
    size_t off = 0;
    const T* src = host.data();
    UploadGenerated<T>(data_, size_, [&](T* p, size_t len) {
      const T* from = src + off;
      ParallelFor(len, [&](size_t lo, size_t hi) { std::memcpy(static_cast<void*>(p + lo), static_cast<const void*>(from + lo), (hi - lo) * sizeof(T)); });
      off += len;
    });
  } else if (size_) {
    CheckHip(prost_hip_memcpy_h2d(data_, host.data(), size_ * sizeof(T), CurrentStream()), "memcpy_h2d");
    CheckHip(prost_hip_stream_synchronize(CurrentStream()), "stream_synchronize");   // host vector may be a temporary
  }
  return *this;
}
template <typename T>
void device_vector<T>::copy_to(std::vector<T>& host) const {
  host.resize(size_);
  if (size_ * sizeof(T) >= ((size_t)4 << 20)) {
    DownloadAs<T, T>(host.data(), data_, size_);        // pinned staging, chunks overlapped with the host copy
  } else if (size_) {
    CheckHip(prost_hip_memcpy_d2h(host.data(), data_, size_ * sizeof(T), CurrentStream()), "memcpy_d2h");
    CheckHip(prost_hip_stream_synchronize(CurrentStream()), "stream_synchronize");
  }
}
template <typename T>
void device_vector<T>::copy_from(const device_vector& o) {
  if (o.size_ != size_) resize(o.size_);
  if (size_) CheckHip(prost_hip_memcpy_d2d(data_, o.data_, size_ * sizeof(T), CurrentStream()), "memcpy_d2d");
}
template class device_vector<float>;
template class device_vector<double>;
template class device_vector<int32_t>;
template class device_vector<int64_t>;
template class device_vector<uint16_t>;

// ---- linspace: num values + a trailing `end` (reference src/common.cu:33-46) ----
template <typename T>
std::list<double> linspace(T start_in, T end_in, int num_in) {
  const double start = static_cast<double>(start_in), end = static_cast<double>(end_in), num = static_cast<double>(num_in);
  const double delta = (end - start) / (num - 1);
  std::list<double> out;
  for (int i = 0; i < num; ++i) out.push_back(start + delta * i);
  out.push_back(end);
  return out;
}
template std::list<double> linspace<double>(double, double, int);
template std::list<double> linspace<float>(float, float, int);
template std::list<double> linspace<size_t>(size_t, size_t, int);
template std::list<double> linspace<int>(int, int, int);

// ---- CSR (n x m) -> CSC by counting sort on the column index (reference src/common.cu:55-82) ----
// Large matrices: the rows are cut into sub-ranges that are counted and scattered on all host cores.  A sub-range of a banded
// matrix (the stencils and warps this is used for) touches a narrow column range, so its private counters are small; a matrix whose
// sub-ranges together span more than four times the columns takes the sequential sort below.  Entries of a column keep the order
// of their rows either way (the sort is stable), so both forms produce the same arrays.
template <typename T>
static bool csr2csc_parallel(int n, int m, int nz, const T* a, const int32_t* col_idx, const int32_t* row_start, T* csc_a, int32_t* row_idx,
                             int32_t* col_start) {
  const size_t parts = ParallelChunks((size_t)n);
  if (nz < (1 << 22) || parts < 2) return false;
  struct Part { size_t r0 = 0, r1 = 0; int32_t cmin = 0, cmax = -1; std::vector<int32_t> cnt; };
  std::vector<Part> part(parts);
  for (size_t t = 0; t < parts; t++) ParallelChunkRange((size_t)n, t, part[t].r0, part[t].r1);
  // (ParallelFor(n, ...) hands out exactly these sub-ranges: the one that starts at `lo` is found by its first row)
  auto part_of = [&](size_t lo) -> Part& { for (Part& p : part) if (p.r0 == lo) return p; return part[0]; };
  ParallelFor((size_t)n, [&](size_t lo, size_t) {
    Part& p = part_of(lo);
    int32_t cmin = m, cmax = -1;
    for (int32_t j = row_start[p.r0]; j < row_start[p.r1]; j++) { cmin = std::min(cmin, col_idx[j]); cmax = std::max(cmax, col_idx[j]); }
    p.cmin = cmin; p.cmax = cmax;
  });
  size_t span = 0;
  for (const Part& p : part) if (p.cmax >= p.cmin) span += (size_t)(p.cmax - p.cmin + 1);
  if (span > 4 * (size_t)m + 1024) return false;
  ParallelFor((size_t)n, [&](size_t lo, size_t) {
    Part& p = part_of(lo);
    if (p.cmax < p.cmin) return;
    p.cnt.assign((size_t)(p.cmax - p.cmin + 1), 0);
    for (int32_t j = row_start[p.r0]; j < row_start[p.r1]; j++) p.cnt[col_idx[j] - p.cmin]++;
  });
  // column starts, and for every sub-range the position of its first entry of every column it touches (cnt is reused for that)
  std::vector<int32_t> run(m + 1, 0);
  for (const Part& p : part) for (size_t k = 0; k < p.cnt.size(); k++) run[p.cmin + k + 1] += p.cnt[k];
  for (int c = 0; c < m; c++) run[c + 1] += run[c];
  for (int c = 0; c <= m; c++) col_start[c] = run[c];
  for (Part& p : part) for (size_t k = 0; k < p.cnt.size(); k++) { const int32_t c = p.cnt[k]; p.cnt[k] = run[p.cmin + k]; run[p.cmin + k] += c; }
  ParallelFor((size_t)n, [&](size_t lo, size_t) {
    Part& p = part_of(lo);
    for (size_t r = p.r0; r < p.r1; r++)
      for (int32_t j = row_start[r]; j < row_start[r + 1]; j++) {
        const int32_t dst = p.cnt[col_idx[j] - p.cmin]++;
        row_idx[dst] = (int32_t)r;
        if (a) csc_a[dst] = a[j];
      }
  });
  (void)nz;
  return true;
}
template <typename T>
void csr2csc(int n, int m, int nz, const T* a, const int32_t* col_idx, const int32_t* row_start, T* csc_a,
             int32_t* row_idx, int32_t* col_start) {
  if (csr2csc_parallel<T>(n, m, nz, a, col_idx, row_start, csc_a, row_idx, col_start)) return;
  std::vector<int32_t> fill(m + 1, 0);
  for (int i = 0; i < nz; i++) fill[col_idx[i] + 1]++;
  for (int c = 0; c < m; c++) fill[c + 1] += fill[c];
  for (int c = 0; c <= m; c++) col_start[c] = fill[c];
  for (int r = 0; r < n; r++)
    for (int32_t j = row_start[r]; j < row_start[r + 1]; j++) {
      const int32_t dst = fill[col_idx[j]]++;
      row_idx[dst] = r;
      if (a) csc_a[dst] = a[j];
    }
}
template void csr2csc<float>(int, int, int, const float*, const int32_t*, const int32_t*, float*, int32_t*, int32_t*);
template void csr2csc<double>(int, int, int, const double*, const int32_t*, const int32_t*, double*, int32_t*, int32_t*);

// ---- glibc rand(): TYPE_3 additive feedback generator r[i] = r[i-3] + r[i-31] ----
GlibcRand::GlibcRand(unsigned seed) : r_(344) {
  if (seed == 0) seed = 1;
  r_[0] = seed;
  for (int i = 1; i < 31; i++) {
    const long long prev = (int32_t)r_[i - 1];
    long long w = 16807 * (prev % 127773) - 2836 * (prev / 127773);
    if (w < 0) w += 2147483647;
    r_[i] = (uint32_t)w;
  }
  for (int i = 31; i < 34; i++) r_[i] = r_[i - 31];
  for (int i = 34; i < 344; i++) r_[i] = r_[i - 31] + r_[i - 3];
}
int32_t GlibcRand::next() {
  const uint32_t v = r_[r_.size() - 31] + r_[r_.size() - 3];
  r_.push_back(v);
  if (r_.size() > 8192) r_.erase(r_.begin(), r_.begin() + 4096);
  return (int32_t)(v >> 1);
}

// ---- jump-ahead for the TYPE_3 additive-feedback generator --------------------------------------------------------------
// r[i] = r[i-31] + r[i-3] (mod 2^32) is LINEAR: the window w_k = (r[k], ..., r[k+30]) satisfies w_{k+1} = M w_k with a
// fixed 31 x 31 matrix over Z / 2^32, so w_{k+s} = M^s w_k.  M^s by repeated squaring costs 31^3 multiplications per bit of
// s -- microseconds -- and lets every host thread start its own sub-range of the stream: the 16.7 M sequential draws of
// normest's start vector (268 M for a 2048 x 2048 x 64 volume) become parallel, bit-identical to std::rand().
namespace {
typedef std::vector<uint32_t> Mat31;                     // row-major 31 x 31
Mat31 MatMul(const Mat31& a, const Mat31& b) {
  Mat31 c(31 * 31, 0);
  for (int i = 0; i < 31; i++)
    for (int k = 0; k < 31; k++) {
      const uint32_t aik = a[i * 31 + k];
      if (!aik) continue;
      for (int j = 0; j < 31; j++) c[i * 31 + j] += aik * b[k * 31 + j];
    }
  return c;
}
Mat31 StepMatrixPower(size_t s) {
  Mat31 m(31 * 31, 0), result(31 * 31, 0);
  for (int i = 0; i < 30; i++) m[i * 31 + i + 1] = 1;     // w'[i] = w[i+1]
  m[30 * 31 + 0] = 1; m[30 * 31 + 28] = 1;                // w'[30] = r[k+31] = r[k] + r[k+28]
  for (int i = 0; i < 31; i++) result[i * 31 + i] = 1;
  for (; s; s >>= 1) { if (s & 1) result = MatMul(m, result); m = MatMul(m, m); }
  return result;
}
}  // namespace

template <typename T>
void GlibcRand::fill_unit(T* out, size_t n) {
  if (n == 0) return;
  // window at the start of the requested range: the last 31 words generated so far
  std::vector<uint32_t> w0(r_.end() - 31, r_.end());
  const size_t chunks = ParallelChunks(n);
  std::vector<std::vector<uint32_t>> start(chunks);
  start[0] = w0;
  if (chunks > 1) {
    size_t b1, e1; ParallelChunkRange(n, 1, b1, e1);
    const Mat31 jump = StepMatrixPower(b1);              // chunk ranges are equally long except possibly the last
    for (size_t c = 1; c < chunks; c++) {
      size_t b, e; ParallelChunkRange(n, c, b, e);
      if (b != c * b1) throw Exception("GlibcRand::fill_unit: unequal chunk ranges.");
      start[c].assign(31, 0);
      for (int i = 0; i < 31; i++) { uint32_t acc = 0; for (int j = 0; j < 31; j++) acc += jump[i * 31 + j] * start[c - 1][j]; start[c][i] = acc; }
    }
  }
  std::vector<uint32_t> last_window(31);
  ParallelFor(n, [&](size_t lo, size_t hi) {
    size_t ci = 0;
    for (size_t c = 0; c < chunks; c++) { size_t b, e; ParallelChunkRange(n, c, b, e); if (b == lo) { ci = c; break; } }
    // a sliding window of 31 words in a small ring: r[i] = r[i-31] + r[i-3]
    uint32_t ring[64];
    for (int i = 0; i < 31; i++) ring[i] = start[ci][i];
    size_t pos = 31;
    for (size_t i = lo; i < hi; i++, pos++) {
      const uint32_t v = ring[(pos - 31) & 63] + ring[(pos - 3) & 63];
      ring[pos & 63] = v;
      out[i] = (T)(int32_t)(v >> 1) / (T)2147483647;
    }
    if (hi == n) for (int i = 0; i < 31; i++) last_window[i] = ring[(pos - 31 + i) & 63];
  });
  r_.assign(last_window.begin(), last_window.end());     // (fewer than 31 new words: the ring still held the older ones)
}
template void GlibcRand::fill_unit<float>(float*, size_t);
template void GlibcRand::fill_unit<double>(double*, size_t);

size_t ParallelChunks(size_t n) {
  const size_t kMinChunk = (size_t)1 << 20;
  size_t hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  const size_t t = std::min<size_t>(std::min<size_t>(16, hw), (n + kMinChunk - 1) / kMinChunk);
  return t < 1 ? 1 : t;
}
void ParallelChunkRange(size_t n, size_t i, size_t& begin, size_t& end) {
  const size_t t = ParallelChunks(n), per = (n + t - 1) / t;
  begin = std::min(n, i * per);
  end = std::min(n, begin + per);
}
void ParallelFor(size_t n, const std::function<void(size_t, size_t)>& fn) {
  const size_t t = ParallelChunks(n);
  if (t <= 1) { fn(0, n); return; }
  std::vector<std::thread> th;
  std::exception_ptr err = nullptr;
  std::mutex mu;
  for (size_t i = 1; i < t; i++) {
    size_t b, e; ParallelChunkRange(n, i, b, e);
    th.emplace_back([&, b, e]() { try { fn(b, e); } catch (...) { std::lock_guard<std::mutex> g(mu); err = std::current_exception(); } });
  }
  size_t b0, e0; ParallelChunkRange(n, 0, b0, e0);
  try { fn(b0, e0); } catch (...) { std::lock_guard<std::mutex> g(mu); err = std::current_exception(); }
  for (auto& x : th) x.join();
  if (err) std::rethrow_exception(err);
}

namespace {
struct DownloadStage {
  static constexpr size_t kBytes = (size_t)32 << 20;
  void* buf[2] = {nullptr, nullptr};
  void* ev[2] = {nullptr, nullptr};
  void ensure() {
    for (int i = 0; i < 2; i++) {
      if (!buf[i] && prost_hip_host_alloc(&buf[i], kBytes) != 0) {
        // no pinned memory to be had (locked-memory limit): a pageable buffer keeps every transfer correct -- the copies
        // are then staged by the runtime and the event waits below still order them -- only the overlap is lost
        buf[i] = std::malloc(kBytes);
        if (!buf[i]) throw Exception("Out of host memory for the transfer staging buffers.");
      }
      if (!ev[i]) CheckHip(prost_hip_event_create(&ev[i]), "event_create");
    }
  }
};
// one stage per device (events belong to the device they were created on); all transfers of the process serialise on one lock
std::mutex g_stage_mu;
std::map<int, DownloadStage> g_stages;
DownloadStage& stage_of_current_device() {
  int dev = 0;
  CheckHip(prost_hip_get_device(&dev), "get_device");
  DownloadStage& st = g_stages[dev];
  st.ensure();
  return st;
}
}  // namespace

template <class D, class T>
void DownloadAs(D* dst, const T* dev, size_t n) {
  if (n == 0) return;
  void* st = CurrentStream();
  std::lock_guard<std::mutex> lock(g_stage_mu);
  DownloadStage& stg = stage_of_current_device();
  const size_t ce = DownloadStage::kBytes / sizeof(T), chunks = (n + ce - 1) / ce;
  auto issue = [&](size_t k) {
    const size_t b = k * ce, len = std::min(ce, n - b);
    CheckHip(prost_hip_memcpy_d2h(stg.buf[k & 1], dev + b, len * sizeof(T), st), "memcpy_d2h");
    CheckHip(prost_hip_event_record(stg.ev[k & 1], st), "event_record");
  };
  issue(0);
  for (size_t k = 0; k < chunks; k++) {
    if (k + 1 < chunks) issue(k + 1);                   // into the buffer whose contents were moved out in the previous round
    CheckHip(prost_hip_event_synchronize(stg.ev[k & 1]), "event_synchronize");
    const size_t b = k * ce, len = std::min(ce, n - b);
    const T* src = static_cast<const T*>(stg.buf[k & 1]);
    D* out = dst + b;
    ParallelFor(len, [&](size_t lo, size_t hi) {
      if (std::is_same<D, T>::value) std::memcpy(static_cast<void*>(out + lo), static_cast<const void*>(src + lo), (hi - lo) * sizeof(T));
      else for (size_t i = lo; i < hi; i++) out[i] = (D)src[i];
    });
  }
}
template <class T>
void UploadGenerated(T* dev, size_t n, const std::function<void(T*, size_t)>& gen) {
  if (n == 0) return;
  void* st = CurrentStream();
  std::lock_guard<std::mutex> lock(g_stage_mu);
  DownloadStage& stg = stage_of_current_device();
  const size_t ce = DownloadStage::kBytes / sizeof(T), chunks = (n + ce - 1) / ce;
  for (size_t k = 0; k < chunks; k++) {
    if (k >= 2) CheckHip(prost_hip_event_synchronize(stg.ev[k & 1]), "event_synchronize");     // the copy that last read this buffer
    const size_t b = k * ce, len = std::min(ce, n - b);
    T* p = static_cast<T*>(stg.buf[k & 1]);
    gen(p, len);
    CheckHip(prost_hip_memcpy_h2d(dev + b, p, len * sizeof(T), st), "memcpy_h2d");
    CheckHip(prost_hip_event_record(stg.ev[k & 1], st), "event_record");
  }
  CheckHip(prost_hip_stream_synchronize(st), "stream_synchronize");
}
template void UploadGenerated<float>(float*, size_t, const std::function<void(float*, size_t)>&);
template void UploadGenerated<double>(double*, size_t, const std::function<void(double*, size_t)>&);
template void UploadGenerated<int32_t>(int32_t*, size_t, const std::function<void(int32_t*, size_t)>&);

template void DownloadAs<float, float>(float*, const float*, size_t);
template void DownloadAs<double, double>(double*, const double*, size_t);
template void DownloadAs<double, float>(double*, const float*, size_t);
template void DownloadAs<int32_t, int32_t>(int32_t*, const int32_t*, size_t);

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
StageTimer::StageTimer(const char* name) : name_(name), t0_(0), on_(std::getenv("PROST_TIMING") != nullptr) {
  if (on_) t0_ = now_ms();
}
StageTimer::~StageTimer() {
  if (!on_) return;
  prost_hip_stream_synchronize(CurrentStream());
  std::fprintf(stderr, "[prost timing] %-40s %9.3f ms\n", name_, now_ms() - t0_);
}

}  // namespace prost
