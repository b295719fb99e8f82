// linop.cpp -- blocks and the block container of the prost host library (calls prost_hip.h only).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <atomic>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "hipapi.hpp"
#include "prost/linop/blocks.hpp"
#include "prost/linop/linearoperator.hpp"

namespace prost {

// ---- Block defaults: non-accumulating = zero fill + accumulate ----
template <typename T>
void Block<T>::EvalLocal(T* rb, T* re, const T* xb, const T* xe) {
  CheckHip(Api<T>::scale(rb, (size_t)(re - rb), 0.0, CurrentStream()), "scale");
  EvalLocalAdd(rb, re, xb, xe);
}
template <typename T>
void Block<T>::EvalAdjointLocal(T* rb, T* re, const T* xb, const T* xe) {
  CheckHip(Api<T>::scale(rb, (size_t)(re - rb), 0.0, CurrentStream()), "scale");
  EvalAdjointLocalAdd(rb, re, xb, xe);
}
template class Block<float>;
template class Block<double>;

// ---- gradient blocks ----
#define GRAD_IMPL(CLS, FWD, ADJ, CS)                                                                                   \
  template <typename T> void CLS<T>::EvalLocalAdd(T* r, T*, const T* x, const T*) { CheckHip(Api<T>::FWD(r, x, nx_, ny_, L_, label_first_, 1, CurrentStream()), #FWD); }          \
  template <typename T> void CLS<T>::EvalAdjointLocalAdd(T* r, T*, const T* x, const T*) { CheckHip(Api<T>::ADJ(r, x, nx_, ny_, L_, label_first_, 1, CurrentStream()), #ADJ); }   \
  template <typename T> void CLS<T>::EvalLocal(T* r, T*, const T* x, const T*) { CheckHip(Api<T>::FWD(r, x, nx_, ny_, L_, label_first_, 0, CurrentStream()), #FWD); }             \
  template <typename T> void CLS<T>::EvalAdjointLocal(T* r, T*, const T* x, const T*) { CheckHip(Api<T>::ADJ(r, x, nx_, ny_, L_, label_first_, 0, CurrentStream()), #ADJ); }      \
  template <typename T> void CLS<T>::row_sums(T* out, T) const { ParallelFor(this->nrows(), [&](size_t b, size_t e) { for (size_t r = b; r < e; r++) out[r] += 2; }); }    \
  template <typename T> void CLS<T>::col_sums(T* out, T) const { ParallelFor(this->ncols(), [&](size_t b, size_t e) { for (size_t c = b; c < e; c++) out[c] += CS; }); }   \
  template class CLS<float>;                                                                                           \
  template class CLS<double>;
GRAD_IMPL(BlockGradient2D, grad2d_fwd, grad2d_adj, 4)
GRAD_IMPL(BlockGradient3D, grad3d_fwd, grad3d_adj, 6)
#undef GRAD_IMPL

// ---- sparse block ----
template <typename T>
BlockSparse<T>* BlockSparse<T>::CreateFromCSC(size_t row, size_t col, int m, int n, int nnz, const std::vector<T>& val,
                                              const std::vector<int32_t>& ptr, const std::vector<int32_t>& ind) {
  return CreateFromCSC(row, col, m, n, nnz, std::vector<T>(val), std::vector<int32_t>(ptr), std::vector<int32_t>(ind));
}
template <typename T>
BlockSparse<T>* BlockSparse<T>::CreateFromCSC(size_t row, size_t col, int m, int n, int nnz, std::vector<T>&& val, std::vector<int32_t>&& ptr,
                                              std::vector<int32_t>&& ind) {
  BlockSparse<T>* b = new BlockSparse<T>(row, col, m, n);
  b->nnz_ = nnz;
  // the CSC arrays of K are the CSR arrays of K^T; K itself in CSR comes from one transposition
  b->host_ind_t_ = std::move(ind); b->host_ptr_t_ = std::move(ptr); b->host_val_t_ = std::move(val);
  b->host_ind_.resize(nnz); b->host_val_.resize(nnz); b->host_ptr_.resize(m + 1);
  csr2csc<T>(n, m, nnz, b->host_val_t_.data(), b->host_ind_t_.data(), b->host_ptr_t_.data(), b->host_val_.data(),
             b->host_ind_.data(), b->host_ptr_.data());
  return b;
}
// Rows of a CSR matrix as repeated (column - row, value) sequences.  Returns false -- the matrix stays CSR -- when it is small (< 256
// rows), when its rows are not made of few sequences (more than a quarter of up to 4096 sampled rows differ; > 4096 sequences in all, or
// a table of > 65536 entries), or on a signature collision.
namespace {
bool g_sparse_patterns = true;
template <typename T>
struct HostRowPatterns { std::vector<uint16_t> ids; std::vector<int32_t> pptr, rel, anchor; std::vector<T> val; };
template <typename T>
bool BuildRowPatterns(size_t nrows, const std::vector<int32_t>& ptr, const std::vector<int32_t>& ind, const std::vector<T>& val, HostRowPatterns<T>& out, bool anchored = false) {
  // anchored: offsets count from the row's FIRST COLUMN instead of from the row number (matrices between different geometries)
  auto base_of = [&](size_t r) -> int32_t { return anchored ? (ptr[r + 1] > ptr[r] ? ind[ptr[r]] : 0) : (int32_t)r; };
  constexpr size_t kMinRows = 256, kMaxPatterns = 4096, kMaxTable = (size_t)1 << 16;
  if (nrows < kMinRows || ptr.size() != nrows + 1) return false;
  const size_t kSample = std::min<size_t>(4096, nrows), kMaxSampled = kSample / 4;
  std::vector<uint64_t> sig(nrows);
  ParallelFor(nrows, [&](size_t lo, size_t hi) {
    for (size_t r = lo; r < hi; r++) {
      uint64_t h = 1469598103934665603ull ^ (uint64_t)(ptr[r + 1] - ptr[r]);
      for (int32_t j = ptr[r]; j < ptr[r + 1]; j++) {
        uint64_t bits = 0;
        std::memcpy(&bits, &val[j], sizeof(T));
        h = (h ^ (uint64_t)(uint32_t)(ind[j] - base_of(r))) * 1099511628211ull; h ^= h >> 29;
        h = (h ^ bits) * 1099511628211ull; h ^= h >> 31;
      }
      sig[r] = h;
    }
  });
  {   // unstructured matrices leave here after a look at 4096 evenly spaced rows
    std::vector<uint64_t> smp(kSample);
    for (size_t i = 0; i < kSample; i++) smp[i] = sig[i * (nrows / kSample)];
    std::sort(smp.begin(), smp.end());
    if ((size_t)(std::unique(smp.begin(), smp.end()) - smp.begin()) > kMaxSampled) return false;
  }
  // distinct signatures and the first row of each (per sub-range, merged), numbered in row order; then every row looks its number
  // up and is compared with the first row of its pattern, entry by entry (equal signatures of different rows: the matrix stays CSR)
  std::unordered_map<uint64_t, size_t> first;
  std::mutex merge;
  std::atomic<bool> ok(true);
  ParallelFor(nrows, [&](size_t lo, size_t hi) {
    std::unordered_map<uint64_t, size_t> local;
    for (size_t r = lo; r < hi && ok.load(std::memory_order_relaxed); r++) {
      if (local.emplace(sig[r], r).second && local.size() > kMaxPatterns) ok = false;
    }
    std::lock_guard<std::mutex> g(merge);
    for (const auto& kv : local) { auto it = first.emplace(kv.first, kv.second).first; if (kv.second < it->second) it->second = kv.second; }
  });
  if (!ok || first.size() > kMaxPatterns) return false;
  std::vector<std::pair<size_t, uint64_t>> order;
  order.reserve(first.size());
  for (const auto& kv : first) order.emplace_back(kv.second, kv.first);
  std::sort(order.begin(), order.end());
  std::unordered_map<uint64_t, uint32_t> number;
  std::vector<size_t> rep(order.size());
  out.pptr.assign(1, 0); out.rel.clear(); out.val.clear();
  for (size_t k = 0; k < order.size(); k++) {
    const size_t r = order[k].first;
    number.emplace(order[k].second, (uint32_t)k);
    rep[k] = r;
    for (int32_t j = ptr[r]; j < ptr[r + 1]; j++) { out.rel.push_back(ind[j] - base_of(r)); out.val.push_back(val[j]); }
    out.pptr.push_back((int32_t)out.rel.size());
    if (out.rel.size() > kMaxTable) return false;
  }
  if (out.rel.empty()) return false;                                  // a matrix without entries: nothing to tabulate
  out.ids.assign((nrows + 3) / 4 * 4, 0);
  ParallelFor(nrows, [&](size_t lo, size_t hi) {
    for (size_t r = lo; r < hi; r++) {
      const uint32_t id = number.find(sig[r])->second;
      const size_t q = rep[id];
      const int32_t len = ptr[r + 1] - ptr[r];
      bool same = len == ptr[q + 1] - ptr[q];
      for (int32_t j = 0; same && j < len; j++)
        same = ind[ptr[r] + j] - base_of(r) == ind[ptr[q] + j] - base_of(q) && std::memcmp(&val[ptr[r] + j], &val[ptr[q] + j], sizeof(T)) == 0;
      if (!same) { ok = false; return; }
      out.ids[r] = (uint16_t)id;
    }
  });
  if (!ok) return false;
  out.anchor.clear();
  if (anchored) {
    // an empty row continues the run of the row above it, so that the 4 rows of a lane keep consecutive anchors across it
    out.anchor.assign((nrows + 3) / 4 * 4, 0);
    for (size_t r = 0; r < nrows; r++) out.anchor[r] = ptr[r + 1] > ptr[r] ? ind[ptr[r]] : (r ? out.anchor[r - 1] + 1 : 0);
  }
  return true;
}
}  // namespace
static bool g_stencil_recognition = true;
template <typename T> void BlockSparse<T>::SetStencilRecognition(bool on) { g_stencil_recognition = on; }

/// Is K, entry for entry, spmat_gradient2d(nx, ny, L) -- rows [0, L n): forward differences along x (-1 at the pixel, +1 one column
/// on, nothing in the last column), rows [L n, 2 L n): along y (-1, +1 one row down, nothing in the last row), n = nx ny, label l in
/// columns [l n, (l + 1) n) (spmat_gradient2d.m:7-14)?  ny is read off row 0, n off the first empty row, then every row is checked.
template <typename T>
void BlockSparse<T>::DetectGradient2D() {
  grad_nx_ = grad_ny_ = grad_L_ = 0;
  const size_t m = this->nrows(), n = this->ncols();
  if (!g_stencil_recognition || m != 2 * n || n < 4 || host_ptr_.size() != m + 1) return;
  if (host_ptr_[1] - host_ptr_[0] != 2 || host_ind_[0] != 0 || host_val_[0] != (T)-1 || host_val_[1] != (T)1) return;
  const size_t ny = (size_t)host_ind_[1];
  if (ny < 2 || ny >= n) return;
  size_t first_empty = n;
  for (size_t r = 0; r < n; r++) if (host_ptr_[r + 1] == host_ptr_[r]) { first_empty = r; break; }
  const size_t img = first_empty + ny;
  if (first_empty == n || img > n || n % img != 0 || img % ny != 0 || img / ny < 2) return;
  std::atomic<bool> ok(true);
  auto row_is = [&](size_t r, size_t c0, bool two) {
    const int32_t b = host_ptr_[r], e = host_ptr_[r + 1];
    if (!two) return e == b;
    return e - b == 2 && (size_t)host_ind_[b] == c0 && host_val_[b] == (T)-1 && host_val_[b + 1] == (T)1;
  };
  ParallelFor(n, [&](size_t lo, size_t hi) {
    for (size_t q = lo; q < hi && ok.load(std::memory_order_relaxed); q++) {
      const size_t p = q % img;
      const bool dx = p < img - ny, dy = p % ny < ny - 1;
      if (!row_is(q, q, dx) || (dx && (size_t)host_ind_[host_ptr_[q] + 1] != q + ny) ||
          !row_is(n + q, q, dy) || (dy && (size_t)host_ind_[host_ptr_[n + q] + 1] != q + 1)) { ok = false; return; }
    }
  });
  if (!ok) return;
  grad_ny_ = ny; grad_nx_ = img / ny; grad_L_ = n / img;
}

template <typename T> void BlockSparse<T>::SetPatternCompression(bool on) { g_sparse_patterns = on; }
template <typename T> bool BlockSparse<T>::pattern_compression() { return g_sparse_patterns; }

template <typename T>
void BlockSparse<T>::Initialize() {
  pat_.on = pat_t_.on = false;
  auto build = [&](RowPatterns& p, size_t nrows, const std::vector<int32_t>& hp, const std::vector<int32_t>& hi, const std::vector<T>& hv) {
    HostRowPatterns<T> h;
    // (offsets from the row number first -- 2 bytes per row; then from the row's first column -- 6 bytes per row, for matrices that map
    // between different geometries: convmtx2 of example_deblurring.m)
    if (!g_sparse_patterns || !(BuildRowPatterns<T>(nrows, hp, hi, hv, h) || BuildRowPatterns<T>(nrows, hp, hi, hv, h, true))) return;
    p.ids = h.ids; p.pptr = h.pptr; p.rel = h.rel; p.val = h.val;
    p.anchor.clear();
    if (!h.anchor.empty()) p.anchor = h.anchor;
    p.count = h.pptr.size() - 1;
    p.on = true;
  };
  DetectGradient2D();
  DetectPointwise();
  build(pat_, this->nrows(), host_ptr_, host_ind_, host_val_);
  build(pat_t_, this->ncols(), host_ptr_t_, host_ind_t_, host_val_t_);
  // the CSR arrays of a product that runs from row patterns stay on the host (a 4096^2 gradient: 0.5 GB of upload each)
  if (!pat_.on) { ind_ = host_ind_; ptr_ = host_ptr_; val_ = host_val_; }
  if (!pat_t_.on) { ind_t_ = host_ind_t_; ptr_t_ = host_ptr_t_; val_t_ = host_val_t_; }
}
/// K couples the channels of one pixel: nrows x (L nrows), row i = L entries at columns i, i + nrows, ..., i + (L - 1) nrows -- the
/// shape of [diag(Ix) diag(Iy)] (a warp / data-term matrix of a multi-channel model).  The two-launch CG rounds of the ADMM backend
/// (prost_hip_cgls_pixel_round) read such a block as a per-pixel table of L values.
template <typename T>
void BlockSparse<T>::DetectPointwise() {
  pointwise_planes_ = 0;
  const size_t m = this->nrows(), n = this->ncols();
  if (m == 0 || n % m != 0 || host_ptr_.size() != m + 1) return;
  const size_t L = n / m;
  if (L < 1 || L > 4 || nnz_ != L * m) return;
  std::atomic<bool> ok(true);
  ParallelFor(m, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi && ok.load(std::memory_order_relaxed); i++) {
      if ((size_t)host_ptr_[i] != i * L || (size_t)host_ptr_[i + 1] != (i + 1) * L) { ok.store(false); return; }
      for (size_t c = 0; c < L; c++)
        if ((size_t)host_ind_[i * L + c] != i + c * m) { ok.store(false); return; }
    }
  });
  if (ok.load()) pointwise_planes_ = L;
}
template <typename T>
void BlockSparse<T>::Release() {
  ind_.clear(); ptr_.clear(); val_.clear(); ind_t_.clear(); ptr_t_.clear(); val_t_.clear();
  for (RowPatterns* p : {&pat_, &pat_t_}) { p->on = false; p->ids.clear(); p->pptr.clear(); p->rel.clear(); p->val.clear(); p->anchor.clear(); }
}
template <typename T>
T BlockSparse<T>::row_sum(size_t row, T alpha) const {
  T sum = 0;
  for (int32_t i = host_ptr_[row]; i < host_ptr_[row + 1]; i++) sum += std::pow(std::abs(host_val_[i]), alpha);
  return sum;
}
template <typename T>
T BlockSparse<T>::col_sum(size_t col, T alpha) const {
  T sum = 0;
  for (int32_t i = host_ptr_t_[col]; i < host_ptr_t_[col + 1]; i++) sum += std::pow(std::abs(host_val_t_[i]), alpha);
  return sum;
}
// |v|^alpha: alpha == 1 (the default scaling, both for rows and for columns: 2 - alpha == 1) needs no pow -- pow(x, 1) == x
template <typename T>
static void csr_abs_pow_sums(T* out, size_t n, const std::vector<int32_t>& ptr, const std::vector<T>& val, T alpha) {
  ParallelFor(n, [&](size_t lo, size_t hi) {
    for (size_t r = lo; r < hi; r++) {
      T sum = 0;
      if (alpha == (T)1) { for (int32_t i = ptr[r]; i < ptr[r + 1]; i++) sum += std::abs(val[i]); }
      else { for (int32_t i = ptr[r]; i < ptr[r + 1]; i++) sum += std::pow(std::abs(val[i]), alpha); }
      out[r] += sum;
    }
  });
}
template <typename T> void BlockSparse<T>::row_sums(T* out, T alpha) const { csr_abs_pow_sums<T>(out, this->nrows(), host_ptr_, host_val_, alpha); }
template <typename T> void BlockSparse<T>::col_sums(T* out, T alpha) const { csr_abs_pow_sums<T>(out, this->ncols(), host_ptr_t_, host_val_t_, alpha); }
template <typename T>
size_t BlockSparse<T>::gpu_mem_amount() const {
  size_t bytes = 0;
  if (!pat_.on) bytes += nnz_ * (sizeof(int32_t) + sizeof(T)) + (this->nrows() + 1) * sizeof(int32_t);
  if (!pat_t_.on) bytes += nnz_ * (sizeof(int32_t) + sizeof(T)) + (this->ncols() + 1) * sizeof(int32_t);
  for (const RowPatterns* p : {&pat_, &pat_t_}) bytes += p->ids.size() * sizeof(uint16_t) + (p->pptr.size() + p->rel.size() + p->anchor.size()) * sizeof(int32_t) + p->val.size() * sizeof(T);
  return bytes;
}
template <typename T>
void BlockSparse<T>::PatternProduct(const RowPatterns& p, T* r, const T* x, size_t rows, int acc) {
  if (p.anchor.size())
    CheckHip(Api<T>::pattern_spmv_anchored(r, x, rows, p.ids.data(), p.anchor.data(), p.pptr.data(), p.rel.data(), p.val.data(), (int)p.pptr.size() - 1, (int)p.rel.size(), acc, CurrentStream()), "pattern_spmv_anchored");
  else
    CheckHip(Api<T>::pattern_spmv(r, x, rows, p.ids.data(), p.pptr.data(), p.rel.data(), p.val.data(), (int)p.pptr.size() - 1, (int)p.rel.size(), acc, CurrentStream()), "pattern_spmv");
}
template <typename T>
void BlockSparse<T>::EvalLocalAdd(T* r, T*, const T* x, const T*) {
  if (!pat_.on && val_.size() != nnz_ && nnz_) throw Exception("BlockSparse used before Initialize().");
  if (pat_.on) { PatternProduct(pat_, r, x, this->nrows(), 1); return; }
  CheckHip(Api<T>::csr_spmv_acc(r, x, this->nrows(), nnz_, val_.data(), ptr_.data(), ind_.data(), CurrentStream()), "csr_spmv_acc");
}
template <typename T>
void BlockSparse<T>::EvalAdjointLocalAdd(T* r, T*, const T* x, const T*) {
  if (!pat_t_.on && val_t_.size() != nnz_ && nnz_) throw Exception("BlockSparse used before Initialize().");
  if (pat_t_.on) { PatternProduct(pat_t_, r, x, this->ncols(), 1); return; }
  CheckHip(Api<T>::csr_spmv_acc(r, x, this->ncols(), nnz_, val_t_.data(), ptr_t_.data(), ind_t_.data(), CurrentStream()), "csr_spmv_acc");
}
template <typename T>
void BlockSparse<T>::EvalLocal(T* r, T*, const T* x, const T*) {
  if (!pat_.on && val_.size() != nnz_ && nnz_) throw Exception("BlockSparse used before Initialize().");
  if (pat_.on) { PatternProduct(pat_, r, x, this->nrows(), 0); return; }
  CheckHip(Api<T>::csr_spmv(r, x, this->nrows(), nnz_, val_.data(), ptr_.data(), ind_.data(), CurrentStream()), "csr_spmv");
}
template <typename T>
void BlockSparse<T>::EvalAdjointLocal(T* r, T*, const T* x, const T*) {
  if (!pat_t_.on && val_t_.size() != nnz_ && nnz_) throw Exception("BlockSparse used before Initialize().");
  if (pat_t_.on) { PatternProduct(pat_t_, r, x, this->ncols(), 0); return; }
  CheckHip(Api<T>::csr_spmv(r, x, this->ncols(), nnz_, val_t_.data(), ptr_t_.data(), ind_t_.data(), CurrentStream()), "csr_spmv");
}
template class BlockSparse<float>;
template class BlockSparse<double>;

// ---- Kronecker blocks ----
template <typename T>
BlockKronSparse<T>* BlockKronSparse<T>::CreateFromCSC(bool id_first, size_t row, size_t col, size_t diaglength, int m, int n, int nnz,
                                                      const std::vector<T>& val, const std::vector<int32_t>& ptr, const std::vector<int32_t>& ind) {
  BlockKronSparse<T>* b = new BlockKronSparse<T>(row, col, (size_t)m * diaglength, (size_t)n * diaglength);
  b->id_first_ = id_first; b->diaglength_ = diaglength; b->mat_nnz_ = nnz; b->mat_nrows_ = m; b->mat_ncols_ = n;
  b->host_ind_t_ = ind; b->host_ptr_t_ = ptr; b->host_val_t_ = std::vector<float>(val.begin(), val.end());     // block_sparse_kron_id.cu:77-79
  b->host_ind_.resize(nnz); b->host_val_.resize(nnz); b->host_ptr_.resize(m + 1);
  csr2csc<float>(n, m, nnz, b->host_val_t_.data(), b->host_ind_t_.data(), b->host_ptr_t_.data(), b->host_val_.data(), b->host_ind_.data(),
                 b->host_ptr_.data());
  return b;
}
template <typename T>
void BlockKronSparse<T>::Initialize() {
  ind_ = host_ind_; ptr_ = host_ptr_; val_ = host_val_;
  ind_t_ = host_ind_t_; ptr_t_ = host_ptr_t_; val_t_ = host_val_t_;
}
template <typename T>
void BlockKronSparse<T>::Release() { ind_.clear(); ptr_.clear(); val_.clear(); ind_t_.clear(); ptr_t_.clear(); val_t_.clear(); }
template <typename T>
T BlockKronSparse<T>::row_sum(size_t row, T alpha) const {
  row = id_first_ ? row % mat_nrows_ : row / diaglength_;
  T sum = 0;
  for (int32_t i = host_ptr_[row]; i < host_ptr_[row + 1]; i++) sum += std::pow(std::abs(host_val_[i]), alpha);
  return sum;
}
template <typename T>
T BlockKronSparse<T>::col_sum(size_t col, T alpha) const {
  col = id_first_ ? col % mat_ncols_ : col / diaglength_;
  T sum = 0;
  for (int32_t i = host_ptr_t_[col]; i < host_ptr_t_[col + 1]; i++) sum += std::pow(std::abs(host_val_t_[i]), alpha);
  return sum;
}
template <typename T>
size_t BlockKronSparse<T>::gpu_mem_amount() const {
  return (host_ind_.size() + host_ind_t_.size() + host_ptr_.size() + host_ptr_t_.size()) * sizeof(int32_t) + (host_val_.size() + host_val_t_.size()) * sizeof(T);
}
template <typename T>
void BlockKronSparse<T>::EvalLocalAdd(T* r, T*, const T* x, const T*) {
  if (ptr_.size() != host_ptr_.size()) throw Exception("BlockKronSparse used before Initialize().");
  if (id_first_) CheckHip(Api<T>::id_kron_sparse_acc(r, x, diaglength_, mat_nrows_, mat_ncols_, val_.data(), ptr_.data(), ind_.data(), CurrentStream()), "id_kron_sparse_acc");
  else CheckHip(Api<T>::sparse_kron_id_acc(r, x, diaglength_, mat_nrows_, val_.data(), ptr_.data(), ind_.data(), CurrentStream()), "sparse_kron_id_acc");
}
template <typename T>
void BlockKronSparse<T>::EvalAdjointLocalAdd(T* r, T*, const T* x, const T*) {
  if (ptr_t_.size() != host_ptr_t_.size()) throw Exception("BlockKronSparse used before Initialize().");
  if (id_first_) CheckHip(Api<T>::id_kron_sparse_acc(r, x, diaglength_, mat_ncols_, mat_nrows_, val_t_.data(), ptr_t_.data(), ind_t_.data(), CurrentStream()), "id_kron_sparse_acc");
  else CheckHip(Api<T>::sparse_kron_id_acc(r, x, diaglength_, mat_ncols_, val_t_.data(), ptr_t_.data(), ind_t_.data(), CurrentStream()), "sparse_kron_id_acc");
}
template <typename T>
void BlockKronSparse<T>::EvalLocal(T* r, T*, const T* x, const T*) {
  if (ptr_.size() != host_ptr_.size()) throw Exception("BlockKronSparse used before Initialize().");
  if (id_first_) CheckHip(Api<T>::id_kron_sparse(r, x, diaglength_, mat_nrows_, mat_ncols_, val_.data(), ptr_.data(), ind_.data(), CurrentStream()), "id_kron_sparse");
  else CheckHip(Api<T>::sparse_kron_id(r, x, diaglength_, mat_nrows_, val_.data(), ptr_.data(), ind_.data(), CurrentStream()), "sparse_kron_id");
}
template <typename T>
void BlockKronSparse<T>::EvalAdjointLocal(T* r, T*, const T* x, const T*) {
  if (ptr_t_.size() != host_ptr_t_.size()) throw Exception("BlockKronSparse used before Initialize().");
  if (id_first_) CheckHip(Api<T>::id_kron_sparse(r, x, diaglength_, mat_ncols_, mat_nrows_, val_t_.data(), ptr_t_.data(), ind_t_.data(), CurrentStream()), "id_kron_sparse");
  else CheckHip(Api<T>::sparse_kron_id(r, x, diaglength_, mat_ncols_, val_t_.data(), ptr_t_.data(), ind_t_.data(), CurrentStream()), "sparse_kron_id");
}
template class BlockKronSparse<float>;
template class BlockKronSparse<double>;

// ---- diags block ----
static bool g_diags_quirk = false;
template <typename T> void BlockDiags<T>::SetReferenceGridQuirk(bool on) { g_diags_quirk = on; }
template <typename T> bool BlockDiags<T>::ReferenceGridQuirk() { return g_diags_quirk; }
template <typename T>
BlockDiags<T>::BlockDiags(size_t row, size_t col, size_t nrows, size_t ncols, size_t ndiags, const std::vector<int64_t>& offsets,
                          const std::vector<T>& factors)
    : Block<T>(row, col, nrows, ncols), ndiags_(ndiags), offsets_(offsets) {
  factors_ = std::vector<float>(factors.begin(), factors.end());
  // ascending offsets: the kernels stop at the first diagonal that leaves the matrix.  Selection
  // by pairwise exchange, which is also the order the reference produces for equal offsets
  // (block_diags.cu:110-118).
  for (size_t i = 0; i < ndiags_; i++)
    for (size_t j = i; j < ndiags_; j++)
      if (offsets_[i] > offsets_[j]) { std::swap(offsets_[i], offsets_[j]); std::swap(factors_[i], factors_[j]); }
}
template <typename T>
void BlockDiags<T>::Initialize() {
  if (ndiags_ >= 1024) throw Exception("Out of constant memory. Too many BlockDiags or too many diagonals.");
  d_offsets_ = offsets_;
  d_factors_ = factors_;
}
template <typename T> void BlockDiags<T>::Release() { d_offsets_.clear(); d_factors_.clear(); }
template <typename T>
T BlockDiags<T>::row_sum(size_t row, T alpha) const {
  T sum = 0;
  for (size_t i = 0; i < ndiags_; i++) {
    const long long col = (long long)row + offsets_[i];
    if (col < 0) continue;
    if ((size_t)col >= this->ncols()) break;
    sum += std::pow(std::abs(factors_[i]), alpha);
  }
  return sum;
}
template <typename T>
T BlockDiags<T>::col_sum(size_t col, T alpha) const {
  T sum = 0;
  const long long sc = (long long)col;
  for (size_t i = 0; i < ndiags_; i++) {
    const long long o = offsets_[i];
    if (o <= sc && (sc - o) < (long long)this->nrows() && (sc - o) >= 0) sum += std::pow(std::abs(factors_[i]), alpha);
    if (o > sc) break;
  }
  return sum;
}
template <typename T>
void BlockDiags<T>::EvalLocalAdd(T* r, T*, const T* x, const T*) {
  if (d_offsets_.size() != ndiags_) throw Exception("BlockDiags used before Initialize().");
  CheckHip(Api<T>::diags_fwd(r, x, this->nrows(), this->ncols(), ndiags_, d_offsets_.data(), d_factors_.data(), CurrentStream()), "diags_fwd");
}
template <typename T>
void BlockDiags<T>::EvalAdjointLocalAdd(T* r, T*, const T* x, const T*) {
  if (d_offsets_.size() != ndiags_) throw Exception("BlockDiags used before Initialize().");
  CheckHip(Api<T>::diags_adj(r, x, this->nrows(), this->ncols(), ndiags_, d_offsets_.data(), d_factors_.data(), g_diags_quirk ? 1 : 0, CurrentStream()), "diags_adj");
}
template class BlockDiags<float>;
template class BlockDiags<double>;
template class BlockZero<float>;
template class BlockZero<double>;

// ---- LinearOperator ----
static bool ranges_partition(std::vector<std::pair<size_t, size_t>> r, size_t total) {
  std::sort(r.begin(), r.end());
  size_t pos = 0;
  for (auto& p : r) { if (p.first != pos) return false; pos = p.first + p.second; }
  return pos == total;
}

template <typename T>
void LinearOperator<T>::InitializeHost() {
  nrows_ = ncols_ = 0;
  bool overlap = false;
  for (size_t i = 0; i < blocks_.size(); i++) {
    const Block<T>& a = *blocks_[i];
    nrows_ = std::max(a.row() + a.nrows(), nrows_);
    ncols_ = std::max(a.col() + a.ncols(), ncols_);
    for (size_t j = i + 1; j < blocks_.size(); j++) {
      const Block<T>& b = *blocks_[j];
      const bool cols_meet = a.col() <= b.col() + b.ncols() - 1 && a.col() + a.ncols() - 1 >= b.col();
      const bool rows_meet = a.row() <= b.row() + b.nrows() - 1 && a.row() + a.nrows() - 1 >= b.row();
      overlap |= cols_meet && rows_meet;
    }
  }
  if (overlap) throw Exception("Blocks are overlapping inside the linear operator. Recheck the indices.");
  std::vector<std::pair<size_t, size_t>> rr, cc;
  for (auto& b : blocks_) { rr.emplace_back(b->row(), b->nrows()); cc.emplace_back(b->col(), b->ncols()); }
  rows_exclusive_ = !blocks_.empty() && ranges_partition(rr, nrows_);
  cols_exclusive_ = !blocks_.empty() && ranges_partition(cc, ncols_);
  // first writers and pure accumulators, in list order (the order LinearOperator::Eval accumulates in): a first writer runs its
  // non-accumulating product (0 + sum: the bits of fill + accumulate), nobody fills
  auto plan = [](const std::vector<std::pair<size_t, size_t>>& r, size_t total, std::vector<char>& out) {
    out.clear();
    std::vector<std::pair<size_t, size_t>> done;                      // disjoint [begin, end) ranges already written, sorted
    std::vector<char> p;
    for (const auto& q : r) {
      const size_t b = q.first, e = q.first + q.second;
      size_t covered = 0;
      for (const auto& d : done) { const size_t lo = std::max(b, d.first), hi = std::min(e, d.second); if (lo < hi) covered += hi - lo; }
      if (covered == 0) { p.push_back(0); done.emplace_back(b, e); }
      else if (covered == e - b) p.push_back(1);
      else return;                                                    // partly new: no plan
      // (merge: keep `done` disjoint)
      std::sort(done.begin(), done.end());
      std::vector<std::pair<size_t, size_t>> m;
      for (const auto& d : done) { if (!m.empty() && d.first <= m.back().second) m.back().second = std::max(m.back().second, d.second); else m.push_back(d); }
      done.swap(m);
    }
    if (done.size() == 1 && done[0].first == 0 && done[0].second == total) out.swap(p);
  };
  plan(rr, nrows_, row_plan_);
  plan(cc, ncols_, col_plan_);
}
template <typename T>
void LinearOperator<T>::Initialize() {
  InitializeHost();
  for (auto& b : blocks_) b->Initialize();
}
template <typename T> void LinearOperator<T>::Release() { for (auto& b : blocks_) b->Release(); }

template <typename T>
void LinearOperator<T>::ApplyBeta(device_vector<T>& result, T beta, bool negate_beta) {
  if (beta == 0) CheckHip(Api<T>::scale(result.data(), result.size(), 0.0, CurrentStream()), "scale");
  else if (beta != 1) CheckHip(Api<T>::scale(result.data(), result.size(), negate_beta ? (double)-beta : (double)beta, CurrentStream()), "scale");
}
template <typename T>
void LinearOperator<T>::Eval(device_vector<T>& result, const device_vector<T>& rhs, T beta) {
  if (beta == 0 && rows_exclusive_ && result.size() == nrows_) {        // every row has exactly one writer: no fill pass
    for (auto& b : blocks_) b->Eval(result.data(), rhs.data());
    return;
  }
  if (beta == 0 && !blocks_.empty() && blocks_[0]->row() == 0 && blocks_[0]->nrows() == nrows_ && result.size() == nrows_) {
    // the first block writes every row: it replaces the fill, the others accumulate in the same order as before
    blocks_[0]->Eval(result.data(), rhs.data());
    for (size_t i = 1; i < blocks_.size(); i++) blocks_[i]->EvalAdd(result.data(), rhs.data());
    return;
  }
  if (beta == 0 && row_plan_.size() == blocks_.size() && !blocks_.empty() && result.size() == nrows_) {
    for (size_t i = 0; i < blocks_.size(); i++) { if (row_plan_[i]) blocks_[i]->EvalAdd(result.data(), rhs.data()); else blocks_[i]->Eval(result.data(), rhs.data()); }
    return;
  }
  ApplyBeta(result, beta, false);
  for (auto& b : blocks_) b->EvalAdd(result.data(), rhs.data());
}
template <typename T>
void LinearOperator<T>::EvalAdjoint(device_vector<T>& result, const device_vector<T>& rhs, T beta) {
  if (beta == 0 && cols_exclusive_ && result.size() == ncols_) {
    for (auto& b : blocks_) b->EvalAdjoint(result.data(), rhs.data());
    return;
  }
  if (beta == 0 && !blocks_.empty() && blocks_[0]->col() == 0 && blocks_[0]->ncols() == ncols_ && result.size() == ncols_) {
    blocks_[0]->EvalAdjoint(result.data(), rhs.data());
    for (size_t i = 1; i < blocks_.size(); i++) blocks_[i]->EvalAdjointAdd(result.data(), rhs.data());
    return;
  }
  if (beta == 0 && col_plan_.size() == blocks_.size() && !blocks_.empty() && result.size() == ncols_) {
    for (size_t i = 0; i < blocks_.size(); i++) { if (col_plan_[i]) blocks_[i]->EvalAdjointAdd(result.data(), rhs.data()); else blocks_[i]->EvalAdjoint(result.data(), rhs.data()); }
    return;
  }
  ApplyBeta(result, beta, false);
  for (auto& b : blocks_) b->EvalAdjointAdd(result.data(), rhs.data());
}
template <typename T>
static double timed_eval(LinearOperator<T>* op, bool adjoint, std::vector<T>& result, const std::vector<T>& rhs) {
  static const int repeats = 5;                     // linearoperator.cu:178
  device_vector<T> d_rhs; d_rhs = rhs;
  device_vector<T> d_res(adjoint ? op->ncols() : op->nrows());
  CheckHip(prost_hip_stream_synchronize(CurrentStream()), "sync");
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < repeats; i++) {
    if (adjoint) op->EvalAdjoint(d_res, d_rhs); else op->Eval(d_res, d_rhs);
    CheckHip(prost_hip_stream_synchronize(CurrentStream()), "sync");
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  d_res.copy_to(result);
  return ms / repeats;
}
template <typename T> double LinearOperator<T>::Eval(std::vector<T>& result, const std::vector<T>& rhs) { return timed_eval(this, false, result, rhs); }
template <typename T> double LinearOperator<T>::EvalAdjoint(std::vector<T>& result, const std::vector<T>& rhs) { return timed_eval(this, true, result, rhs); }

template <typename T>
T LinearOperator<T>::row_sum(size_t row, T alpha) const {
  T sum = 0;
  for (auto& b : blocks_) { if (row < b->row() || row >= b->row() + b->nrows()) continue; sum += b->row_sum(row - b->row(), alpha); }
  return sum;
}
template <typename T>
T LinearOperator<T>::col_sum(size_t col, T alpha) const {
  T sum = 0;
  for (auto& b : blocks_) { if (col < b->col() || col >= b->col() + b->ncols()) continue; sum += b->col_sum(col - b->col(), alpha); }
  return sum;
}
template <typename T>
void LinearOperator<T>::row_sums(std::vector<T>& out, T alpha) const {
  ParallelFor(out.size(), [&](size_t b, size_t e) { std::fill(out.begin() + b, out.begin() + e, (T)0); });
  for (auto& b : blocks_) { if (b->row() + b->nrows() <= out.size()) b->row_sums(out.data() + b->row(), alpha); }
}
template <typename T>
void LinearOperator<T>::col_sums(std::vector<T>& out, T alpha) const {
  ParallelFor(out.size(), [&](size_t b, size_t e) { std::fill(out.begin() + b, out.begin() + e, (T)0); });
  for (auto& b : blocks_) { if (b->col() + b->ncols() <= out.size()) b->col_sums(out.data() + b->col(), alpha); }
}
template <typename T>
size_t LinearOperator<T>::gpu_mem_amount() const { size_t m = 0; for (auto& b : blocks_) m += b->gpu_mem_amount(); return m; }
template class LinearOperator<float>;
template class LinearOperator<double>;

// ---- DualLinearOperator: -K^T (dual_linearoperator.cu:39-80) ----
static bool g_negate_quirk = false;
template <typename T> void DualLinearOperator<T>::SetReferenceNegateQuirk(bool on) { g_negate_quirk = on; }
template <typename T>
void DualLinearOperator<T>::Eval(device_vector<T>& result, const device_vector<T>& rhs, T beta) {
  this->ApplyBeta(result, beta, true);
  for (auto& b : child_->blocks_) b->EvalAdjointAdd(result.data(), rhs.data());
  CheckHip(negate(result.data(), result.size(), g_negate_quirk, CurrentStream()), "negate");
}
template <typename T>
void DualLinearOperator<T>::EvalAdjoint(device_vector<T>& result, const device_vector<T>& rhs, T beta) {
  this->ApplyBeta(result, beta, true);
  for (auto& b : child_->blocks_) b->EvalAdd(result.data(), rhs.data());
  CheckHip(negate(result.data(), result.size(), g_negate_quirk, CurrentStream()), "negate");
}
template class DualLinearOperator<float>;
template class DualLinearOperator<double>;

}  // namespace prost
