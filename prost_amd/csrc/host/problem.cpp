// problem.cpp -- problem assembly, coverage checks, preconditioners and normest
// (behaviour of the reference's src/problem.cu, written against the MI355X host layer).
#include <algorithm>
#include <cmath>
#include <iostream>
#include <memory>
#include <sstream>

#include "hipapi.hpp"
#include "prost/problem.hpp"
#include "prost/prox/proxes.hpp"

namespace prost {

template <typename T>
static std::vector<shared_ptr<Prox<T>>> sorted_by_index(const std::vector<shared_ptr<Prox<T>>>& l) {
  std::vector<shared_ptr<Prox<T>>> s = l;
  std::sort(s.begin(), s.end(), [](const shared_ptr<Prox<T>>& a, const shared_ptr<Prox<T>>& b) { return a->index() < b->index(); });
  return s;
}

/// throws unless the prox ranges tile [0, n) exactly (messages as problem.cu:48-89)
template <typename T>
static void CheckDomainProx(const typename Problem<T>::ProxList& proxs, size_t n, const std::string& name) {
  if (proxs.empty()) return;
  auto s = sorted_by_index<T>(proxs);
  for (size_t i = 0; i + 1 < s.size(); i++)
    if (s[i]->end() != s[i + 1]->index() - 1) {
      std::stringstream ss;
      ss << name << " (CheckDomainProx): Prox operators are overlapping: [" << s[i]->index() << ", " << s[i]->end() << "] and ["
         << s[i + 1]->index() << ", " << s[i + 1]->end() << "]." << std::endl;
      throw Exception(ss.str());
    }
  const auto& last = s.back();
  if (last->end() != n - 1) {
    std::stringstream ss;
    ss << name << (last->end() < n - 1 ? " (CheckDomainProx): Last prox operator ends too early: ["
                                       : " (CheckDomainProx): Last prox operator ends after the domain: [");
    ss << last->index() << ", " << last->end() << "], end = " << n - 1 << "." << std::endl;
    throw Exception(ss.str());
  }
}

/// fills uncovered ranges of [0, n) with identity proxes (problem.cu:93-158)
template <typename T>
static void AddZeroProx(typename Problem<T>::ProxList& proxs, size_t n, const std::string& name) {
  if (proxs.empty()) return;
  auto s = sorted_by_index<T>(proxs);
  if (s[0]->index() > 0) proxs.push_back(shared_ptr<Prox<T>>(new ProxZero<T>(0, s[0]->index())));
  for (size_t i = 0; i + 1 < s.size(); i++)
    if (s[i]->end() < s[i + 1]->index() - 1)
      proxs.push_back(shared_ptr<Prox<T>>(new ProxZero<T>(s[i]->end() + 1, s[i + 1]->index() - s[i]->end() - 1)));
  const auto& last = s.back();
  if (last->end() != n - 1) {
    if (last->end() < n - 1) proxs.push_back(shared_ptr<Prox<T>>(new ProxZero<T>(last->end() + 1, (n - 1) - last->end())));
    else {
      std::stringstream ss;
      ss << name << " (AddZeroProx): Last prox operator ends after the domain: [" << last->index() << ", " << last->end()
         << "], end = " << n - 1 << "." << std::endl;
      throw Exception(ss.str());
    }
  }
}

template <typename T>
Problem<T>::Problem() : nrows_(0), ncols_(0), linop_(new LinearOperator<T>()), scaling_type_(kScalingAlpha), scaling_alpha_(1), host_initialized_(false) {}

template <typename T> void Problem<T>::AddBlock(shared_ptr<Block<T>> block) { linop_->AddBlock(block); }

template <typename T>
void Problem<T>::SetScalingCustom(const std::vector<T>& left, const std::vector<T>& right) {
  scaling_type_ = kScalingCustom;
  scaling_left_host_.resize(left.size());
  scaling_right_host_.resize(right.size());
  for (size_t i = 0; i < left.size(); i++) scaling_left_host_[i] = left[i] * left[i];
  for (size_t i = 0; i < right.size(); i++) scaling_right_host_[i] = right[i] * right[i];
}

/// out[i] = 1 / sums[i] where sums[i] > 0, else the last value written before it (`carry` before the first entry);
/// returns the value carried out of the sweep.  Same results as the sequential sweep of problem.cu:262-287.
template <typename T>
static T CarriedReciprocals(const std::vector<T>& sums, std::vector<T>& out, T carry) {
  const size_t n = sums.size(), chunks = ParallelChunks(n);
  std::vector<T> last(chunks);            // reciprocal of the last positive sum of each sub-range (or 0: none)
  ParallelFor(n, [&](size_t b, size_t e) {
    size_t ci = 0;
    for (size_t i = 0; i < chunks; i++) { size_t cb, ce; ParallelChunkRange(n, i, cb, ce); if (cb == b) { ci = i; break; } }
    T v = 0;
    for (size_t i = e; i > b; i--) if (sums[i - 1] > 0) { v = (T)(1. / (double)sums[i - 1]); break; }
    last[ci] = v;
  });
  std::vector<T> carry_in(chunks);
  for (size_t i = 0; i < chunks; i++) { carry_in[i] = carry; if (last[i] > 0) carry = last[i]; }
  ParallelFor(n, [&](size_t b, size_t e) {
    size_t ci = 0;
    for (size_t i = 0; i < chunks; i++) { size_t cb, ce; ParallelChunkRange(n, i, cb, ce); if (cb == b) { ci = i; break; } }
    T value = carry_in[ci], last_sum = 0;
    for (size_t r = b; r < e; r++) {
      if (sums[r] > 0 && sums[r] != last_sum) { last_sum = sums[r]; value = (T)(1. / (double)sums[r]); }
      out[r] = value;
    }
  });
  return carry;
}

template <typename T>
void Problem<T>::InitializeHost() {
  if (host_initialized_) return;
  StageTimer timer("Problem::InitializeHost");
  linop_->InitializeHost();
  if (linop_->nrows() != nrows_ || linop_->ncols() != ncols_)
    std::cout << "Size of linear operator (ncols=" << linop_->ncols() << ", nrows=" << linop_->nrows()
              << ") doesn't match size of variables. There might be some unnecessary variables in the problem.\n";
  if (prox_f_.empty() && prox_fstar_.empty()) throw Exception("No proximal operator for f or fstar specified.");
  if (prox_g_.empty() && prox_gstar_.empty()) throw Exception("No proximal operator for g or gstar specified.");
  if (!prox_f_.empty() && !prox_fstar_.empty()) throw Exception("Proximal operator for f AND fstar specified. Only set one!");
  if (!prox_g_.empty() && !prox_gstar_.empty()) throw Exception("Proximal operator for g AND gstar specified. Only set one!");

  if (!prox_f_.empty()) AddZeroProx<T>(prox_f_, nrows_, "prox_f");
  if (!prox_g_.empty()) AddZeroProx<T>(prox_g_, ncols_, "prox_g");
  if (!prox_fstar_.empty()) AddZeroProx<T>(prox_fstar_, nrows_, "prox_fstar");
  if (!prox_gstar_.empty()) AddZeroProx<T>(prox_gstar_, ncols_, "prox_gstar");
  CheckDomainProx<T>(prox_g_, ncols_, "prox_g");
  CheckDomainProx<T>(prox_f_, nrows_, "prox_f");
  CheckDomainProx<T>(prox_gstar_, ncols_, "prox_gstar");
  CheckDomainProx<T>(prox_fstar_, nrows_, "prox_fstar");

  left_uniform_ = right_uniform_ = false;
  T urow = 0, ucol = 0;
  if (scaling_type_ == kScalingAlpha && linop_->uniform_sums(scaling_alpha_, (T)(2. - (double)scaling_alpha_), urow, ucol) && urow > 0 && ucol > 0) {
    // ONE stencil block with constant row / column sums (gradient2d / 3d): Sigma and Tau are one value each -- the reciprocal
    // sweep of problem.cu:262-287 would write it into every entry.  The averaging over separable groups (problem.cu:503-536)
    // maps a constant vector to a constant vector when every prox of the list agrees on the mean; then nothing is swept,
    // stored or uploaded per entry (2048 x 2048 x 64: 1.07e9 entries).  Otherwise fall through to the vectors.
    T lv = (T)(1. / (double)urow), rv = (T)(1. / (double)ucol);
    auto uniform_average = [](T& v, const ProxList& list) {
      bool first = true; T out = v;
      for (auto& p : list) {
        T pv = v;
        if (!p->diagsteps() && !p->average_uniform(pv)) return false;
        if (first) { out = pv; first = false; } else if (pv != out) return false;
      }
      v = out;
      return true;
    };
    if (uniform_average(rv, prox_g_.empty() ? prox_gstar_ : prox_g_) && uniform_average(lv, prox_f_.empty() ? prox_fstar_ : prox_f_)) {
      left_uniform_ = right_uniform_ = true;
      left_value_ = lv; right_value_ = rv;
      scaling_left_host_.clear(); scaling_right_host_.clear();
      host_initialized_ = true;
      return;
    }
  }
  if (scaling_type_ == kScalingAlpha) {
    // Sigma_i = 1 / sum_j |K_ij|^alpha ; Tau_j = 1 / sum_i |K_ij|^(2-alpha).  An all-zero row or
    // column inherits the last positive value seen, and that carry runs from the row sweep on
    // into the column sweep (problem.cu:262-287).
    // the sums are accumulated straight into the preconditioner vectors and inverted in place (at 10^9 entries every extra
    // host vector is seconds of page faults)
    StageTimer t_pre("  preconditioner sums");
    scaling_left_host_.resize(nrows_);
    scaling_right_host_.resize(ncols_);
    linop_->row_sums(scaling_left_host_, scaling_alpha_);
    // (the reciprocal is only re-evaluated when the sum changes: stencil operators repeat one value 10^7 times; the sweep
    // runs on several host threads -- a sub-range starts from the value the sweep carries into it, found in a first pass)
    T value = 1;
    value = CarriedReciprocals(scaling_left_host_, scaling_left_host_, value);
    linop_->col_sums(scaling_right_host_, (T)(2. - (double)scaling_alpha_));
    value = CarriedReciprocals(scaling_right_host_, scaling_right_host_, value);
  } else if (scaling_type_ == kScalingIdentity) {
    scaling_left_host_.assign(nrows_, 1);
    scaling_right_host_.assign(ncols_, 1);
  } else {
    if (scaling_left_host_.size() != nrows_ || scaling_right_host_.size() != ncols_)
      throw Exception("Preconditioners/diagonal scaling vectors do not fit the size of linear operator.");
  }
  StageTimer t_avg("  AveragePreconditioners");
  AveragePreconditioners(scaling_right_host_, prox_g_.empty() ? prox_gstar_ : prox_g_);
  AveragePreconditioners(scaling_left_host_, prox_f_.empty() ? prox_fstar_ : prox_f_);
  host_initialized_ = true;
}

template <typename T>
void Problem<T>::Initialize() {
  InitializeHost();
  StageTimer timer("Problem::Initialize uploads");
  for (auto& b : linop_->blocks()) b->Initialize();
  for (auto& p : prox_f_) p->Initialize();
  for (auto& p : prox_fstar_) p->Initialize();
  for (auto& p : prox_g_) p->Initialize();
  for (auto& p : prox_gstar_) p->Initialize();
  if (left_uniform_ && right_uniform_) {            // constants: filled on the device, nothing crosses PCIe
    scaling_left_.resize(nrows_); scaling_right_.resize(ncols_);
    CheckHip(Api<T>::fill(scaling_left_.data(), (double)left_value_, nrows_, CurrentStream()), "fill");
    CheckHip(Api<T>::fill(scaling_right_.data(), (double)right_value_, ncols_, CurrentStream()), "fill");
  } else {
    scaling_left_ = scaling_left_host_;
    scaling_right_ = scaling_right_host_;
  }
  dual_linop_ = shared_ptr<LinearOperator<T>>(new DualLinearOperator<T>(linop_));
}

template <typename T>
void Problem<T>::Release() {
  linop_->Release();
  for (auto& p : prox_f_) p->Release();
  for (auto& p : prox_fstar_) p->Release();
  for (auto& p : prox_g_) p->Release();
  for (auto& p : prox_gstar_) p->Release();
  scaling_left_.clear();
  scaling_right_.clear();
}

template <typename T>
size_t Problem<T>::gpu_mem_amount() const {
  size_t mem = 0;
  for (auto& p : prox_f_) mem += p->gpu_mem_amount();
  for (auto& p : prox_g_) mem += p->gpu_mem_amount();
  for (auto& p : prox_fstar_) mem += p->gpu_mem_amount();
  for (auto& p : prox_gstar_) mem += p->gpu_mem_amount();
  mem += linop_->gpu_mem_amount();
  mem += sizeof(T) * (nrows() + ncols());
  return mem;
}

/// a prox that cannot take a diagonal step size gets the mean preconditioner over each of its
/// separable groups (problem.cu:503-536)
template <typename T>
void Problem<T>::AveragePreconditioners(std::vector<T>& precond, const ProxList& prox) {
  for (auto& p : prox) if (!p->diagsteps()) p->average_preconditioner(precond);
}

template <typename T>
void Problem<T>::Dualize() {
  prox_g_.swap(prox_fstar_);
  prox_gstar_.swap(prox_f_);
  std::swap(nrows_, ncols_);
  std::swap(linop_, dual_linop_);
  scaling_left_.swap(scaling_right_);
  std::swap(scaling_left_host_, scaling_right_host_);
  std::swap(left_uniform_, right_uniform_);
  std::swap(left_value_, right_value_);
}

template <typename T>
void Problem<T>::MaterializeHost() const {
  if (left_uniform_ && scaling_left_host_.size() != nrows_) {
    scaling_left_host_.resize(nrows_);
    ParallelFor(nrows_, [&](size_t b, size_t e) { std::fill(scaling_left_host_.begin() + b, scaling_left_host_.begin() + e, left_value_); });
  }
  if (right_uniform_ && scaling_right_host_.size() != ncols_) {
    scaling_right_host_.resize(ncols_);
    ParallelFor(ncols_, [&](size_t b, size_t e) { std::fill(scaling_right_host_.begin() + b, scaling_right_host_.begin() + e, right_value_); });
  }
}

/// |Sigma^(1/2) K Tau^(1/2)| by power iteration on the device (problem.cu:429-500).  The start
/// vector is the std::rand() stream of a fresh process (srand is never called by the reference).
template <typename T>
T Problem<T>::normest(T tol, int max_iters) {
  const size_t n = ncols(), m = nrows();
  // ONE gradient block under constant preconditioners (the ROF / TV problems): the whole round is one stencil kernel and K x
  // is never stored (prost_hip_normest_grad_round: 2 instead of 23 values per voxel through HBM per round)
  BlockDesc bd;
  bool grad_round = dynamic_cast<DualLinearOperator<T>*>(linop_.get()) == nullptr && linop_->blocks().size() == 1 && left_uniform_ && right_uniform_;
  if (grad_round) {
    auto blk = linop_->blocks()[0];
    grad_round = blk->describe(bd) && !bd.label_first && (bd.kind == BlockDesc::kGradient2D || bd.kind == BlockDesc::kGradient3D) &&
                 blk->row() == 0 && blk->col() == 0 && blk->nrows() == m && blk->ncols() == n &&
                 n == bd.nx * bd.ny * bd.L && m == (bd.kind == BlockDesc::kGradient3D ? 3 : 2) * n;
  }
  device_vector<T> x(n), x_temp(n), Ax_temp(grad_round ? 0 : m);
  {
    StageTimer t_rng("  normest: start vector (glibc rand stream)");
    // generated piecewise into pinned staging buffers, every piece on its way to the device while the next one is drawn
    GlibcRand rng(1);
    UploadGenerated<T>(x.data(), n, [&](T* p, size_t len) { rng.fill_unit(p, len); });      // x[i] = (T)rand() / (T)RAND_MAX (problem.cu:441-444)
  }
  // Per round the reference runs four scaling passes, two nrm2 (each with a blocking read-back) and a divide around K
  // and K^T; here three fused passes (prost_hip_normest_stage_*: same expressions and roundings), the two norms land in
  // pinned host memory and ONE synchronisation per round reads them.
  void* ws = nullptr;
  double* out_host = nullptr;
  CheckHip(prost_hip_malloc(&ws, prost_hip_cgls_workspace_bytes()), "malloc");
  CheckHip(prost_hip_host_alloc((void**)&out_host, 2 * (size_t)std::max(max_iters, 1) * sizeof(double)), "host_alloc");
  StageTimer t_rounds("  normest: power iteration");
  prost_hip_normest_desc d;
  d.workspace = ws; d.x = x.data(); d.x_temp = x_temp.data(); d.ax = Ax_temp.data();
  d.sigma = scaling_left_.data(); d.tau = scaling_right_.data(); d.m = m; d.n = n; d.norm_x = 0; d.out = out_host; d.norm_x_from = nullptr;
  // Only the norm leaves this function, so rounds beyond the one at which the reference's loop stops are harmless: the rounds
  // are queued in batches without a host round trip in between -- round i writes its two norms to slot i of a pinned array,
  // round i + 1 reads its divisor from there (norm_x_from) -- and the stopping test of problem.cu:493-496 is replayed on the
  // host over the recorded norms, in order.
  T norm = 0, norm_prev;
  bool stop = false;
  const int kBatch = 20;
  try {
    for (int i0 = 0; i0 < max_iters && !stop; i0 += kBatch) {
      const int i1 = std::min(max_iters, i0 + kBatch);
      for (int i = i0; i < i1; i++) {
        d.out = out_host + 2 * i;
        d.norm_x_from = i > 0 ? out_host + 2 * (i - 1) + 1 : nullptr;
        if (grad_round) {
          prost_hip_normest_grad_desc g;
          g.is3d = bd.kind == BlockDesc::kGradient3D ? 1 : 0; g.nx = bd.nx; g.ny = bd.ny; g.L = bd.L;
          g.x_in = (i & 1) ? x_temp.data() : x.data(); g.x_out = (i & 1) ? x.data() : x_temp.data();     // ping-pong (stencil)
          g.tau = (double)right_value_; g.sigma = (double)left_value_;
          g.norm_x_from = d.norm_x_from; g.out = d.out; g.workspace = ws;
          CheckHip(Api<T>::normest_grad_round(&g, CurrentStream()), "normest_grad_round");
          continue;
        }
        CheckHip(Api<T>::normest_stage(PROST_NORMEST_A, &d, CurrentStream()), "normest_stage");
        linop_->Eval(Ax_temp, x_temp);
        CheckHip(Api<T>::normest_stage(PROST_NORMEST_B, &d, CurrentStream()), "normest_stage");
        linop_->EvalAdjoint(x_temp, Ax_temp);
        CheckHip(Api<T>::normest_stage(PROST_NORMEST_C, &d, CurrentStream()), "normest_stage");
      }
      CheckHip(prost_hip_stream_synchronize(CurrentStream()), "sync");
      for (int i = i0; i < i1 && !stop; i++) {
        norm_prev = norm;
        // the reference reduces in T with an unspecified tree order; here in double, narrowed to T
        const T norm_Ax = (T)out_host[2 * i], norm_x = (T)out_host[2 * i + 1];
        norm = norm_x / norm_Ax;
        if (std::abs(norm_prev - norm) < tol * norm) stop = true;
      }
    }
  } catch (...) { prost_hip_free(ws); prost_hip_host_free(out_host); throw; }
  prost_hip_free(ws);
  prost_hip_host_free(out_host);
  return norm;
}

template class Problem<float>;
template class Problem<double>;

}  // namespace prost
