// backend_admm.cpp -- graph-projection ADMM with CGLS inner solves on the MI355X kernels
// (behaviour of the reference's src/backend/backend_admm.cu and include/prost/cgls.hpp; the
// cuBLAS nrm2/axpy and thrust functors are the prost_hip_nrm2 / _axpy / _admm_elem kernels).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <cmath>
#include <limits>

#include "hipapi.hpp"
#include "prost/backend/backend_admm.hpp"
#include "prost/prox/proxes.hpp"

namespace prost {

template <typename T> BackendADMM<T>::~BackendADMM() { Release(); }

template <typename T>
void BackendADMM<T>::Initialize() {
  const size_t m = this->problem_->nrows(), n = this->problem_->ncols(), l = std::max(m, n);
  x_half_.resize(n); x_proj_.resize(n); x_dual_.resize(n);
  z_half_.resize(m); z_proj_.resize(m); z_dual_.resize(m);
  temp1_.resize(n);
  temp2_.resize(l);      // the reference scales n entries into this m-vector (:586-590); sized max(m,n) here
  temp3_.resize(l);
  tmp_n_.resize(n); tmp_m_.resize(m);
  prox_g_.clear(); prox_f_.clear();
  if (this->problem_->prox_g().empty()) {
    if (this->problem_->prox_gstar().empty()) throw Exception("Neither prox_g nor prox_gstar specified.");
    for (auto& p : this->problem_->prox_gstar()) { auto mo = std::make_shared<ProxMoreau<T>>(p); mo->Initialize(); prox_g_.push_back(mo); }
  } else prox_g_ = this->problem_->prox_g();
  if (this->problem_->prox_f().empty()) {
    if (this->problem_->prox_fstar().empty()) throw Exception("Neither prox_f nor prox_fstar specified.");
    for (auto& p : this->problem_->prox_fstar()) { auto mo = std::make_shared<ProxMoreau<T>>(p); mo->Initialize(); prox_f_.push_back(mo); }
  } else prox_f_ = this->problem_->prox_f();
  delta_ = opts_.arb_delta;
  rho_ = (T)opts_.rho0;
  iteration_ = 0;
  arb_u_ = arb_l_ = 0;
  // the reference leaves the residual members uninitialised; zero them
  this->primal_var_norm_ = this->dual_var_norm_ = this->primal_residual_ = this->dual_residual_ = 0;
  CheckHip(prost_hip_malloc((void**)&scal_dev_, 8 * sizeof(double)), "malloc");
  CheckHip(prost_hip_host_alloc((void**)&scal_host_, 8 * sizeof(double)), "host_alloc");
  CheckHip(prost_hip_malloc(&workspace_, prost_hip_reduce_workspace_bytes()), "malloc");
  // one scalar record per CG round + the initial one (prost_hip_cgls_round_*); the staged rounds use record 0 only
  const size_t records = (size_t)std::max(opts_.cg_max_iter, 0) + 2;
  CheckHip(prost_hip_malloc(&cg_state_, records * prost_hip_cgls_state_bytes()), "malloc");
  CheckHip(prost_hip_memset(cg_state_, 0, records * prost_hip_cgls_state_bytes(), CurrentStream()), "memset");
  fused_rounds_ = false;
  pixel_rounds_ = false;
  fused_op_.nblocks = 0;
  cg_result_index_ = 0;
  if (opts_.device_cg && opts_.fused_rounds && !opts_.cg_graph) DescribeOperator();
  CheckHip(prost_hip_malloc(&cg_workspace_, prost_hip_cgls_workspace_bytes()), "malloc");
  CheckHip(prost_hip_host_alloc((void**)&cg_done_host_, sizeof(int)), "host_alloc");
  *cg_done_host_ = 0;
  cg_epoch_ = 0;
  cg_iters_valid_ = true;
  if (opts_.cg_graph) {
    CheckHip(prost_hip_stream_create(&cg_stream_), "stream_create");
    CheckHip(prost_hip_event_create(&cg_ev_[0]), "event_create");
    CheckHip(prost_hip_event_create(&cg_ev_[1]), "event_create");
  }
}

template <typename T>
void BackendADMM<T>::Release() {
  if (scal_dev_) { prost_hip_free(scal_dev_); scal_dev_ = nullptr; }
  if (scal_host_) { prost_hip_host_free(scal_host_); scal_host_ = nullptr; }
  if (workspace_) { prost_hip_free(workspace_); workspace_ = nullptr; }
  if (cg_graph_) { prost_hip_graph_destroy(cg_graph_); cg_graph_ = nullptr; }
  if (cg_stream_) { prost_hip_stream_synchronize(cg_stream_); prost_hip_stream_destroy(cg_stream_); cg_stream_ = nullptr; }
  for (void*& e : cg_ev_) if (e) { prost_hip_event_destroy(e); e = nullptr; }
  if (cg_state_) { prost_hip_free(cg_state_); cg_state_ = nullptr; }
  if (cg_workspace_) { prost_hip_free(cg_workspace_); cg_workspace_ = nullptr; }
  if (cg_done_host_) { prost_hip_host_free(cg_done_host_); cg_done_host_ = nullptr; }
  for (void* e : ev_) prost_hip_event_destroy(e);
  ev_.clear(); ev_used_ = 0;
  x_half_.clear(); z_half_.clear(); x_proj_.clear(); z_proj_.clear(); x_dual_.clear(); z_dual_.clear(); temp1_.clear(); temp2_.clear(); temp3_.clear(); tmp_n_.clear(); tmp_m_.clear(); cg_p_alt_.clear(); cg_r_alt_.clear();
}

template <typename T>
static void elem(int op, T* o, const T* a, const T* b, const T* c, const T* d, double alpha, double beta, size_t n) {
  CheckHip(Api<T>::admm_elem(op, o, a, b, c, d, alpha, beta, n, CurrentStream()), "admm_elem");
}

template <typename T>
void BackendADMM<T>::Gemv(char op, T alpha, const device_vector<T>& x, T beta, device_vector<T>& y) {
  const device_vector<T>& Sl = this->problem_->scaling_left();
  const device_vector<T>& Tr = this->problem_->scaling_right();
  const size_t m = this->problem_->nrows(), n = this->problem_->ncols();
  if (op == 'n') {
    elem<T>(PROST_ADMM_GEMV1, temp3_.data(), Tr.data(), x.data(), nullptr, nullptr, 0, 0, n);          // temp = Tau^(1/2) x
    elem<T>(PROST_ADMM_GEMV2, y.data(), Sl.data(), y.data(), nullptr, nullptr, alpha, beta, m);        // y = beta/(alpha Sigma^(1/2)) y
    this->problem_->linop()->Eval(y, temp3_, 1);                                                       // y += K temp
    elem<T>(PROST_ADMM_GEMV3, y.data(), Sl.data(), y.data(), nullptr, nullptr, alpha, 0, m);           // y = alpha Sigma^(1/2) y
  } else {
    elem<T>(PROST_ADMM_GEMV1, temp3_.data(), Sl.data(), x.data(), nullptr, nullptr, 0, 0, m);
    elem<T>(PROST_ADMM_GEMV2, y.data(), Tr.data(), y.data(), nullptr, nullptr, alpha, beta, n);
    this->problem_->linop()->EvalAdjoint(y, temp3_, 1);
    elem<T>(PROST_ADMM_GEMV3, y.data(), Tr.data(), y.data(), nullptr, nullptr, alpha, 0, n);
  }
}

template <typename T>
double BackendADMM<T>::Nrm2(const device_vector<T>& v, size_t n) {
  CheckHip(Api<T>::nrm2(scal_dev_, v.data(), n, workspace_, CurrentStream()), "nrm2");
  CheckHip(prost_hip_memcpy_d2h(scal_host_, scal_dev_, 2 * sizeof(double), CurrentStream()), "memcpy_d2h");
  CheckHip(prost_hip_stream_synchronize(CurrentStream()), "stream_synchronize");
  return scal_host_[0];
}

/// CGLS for min |Ax - b|^2 + shift |x|^2 with A = Sigma^(1/2) K Tau^(1/2) (cgls.hpp:222-371)
template <typename T>
int BackendADMM<T>::Cgls(const device_vector<T>& b, device_vector<T>& x, double shift, double tol, int maxit, device_vector<T>& p,
                         device_vector<T>& q, device_vector<T>& r, device_vector<T>& s, int& iterations) {
  const size_t m = this->problem_->nrows(), n = this->problem_->ncols();
  void* st = CurrentStream();
  const T kNegShift = (T)(-shift);
  const double kEps = std::numeric_limits<T>::epsilon();
  int k = 0, flag = 0, indefinite = 0;
  CheckHip(prost_hip_memcpy_d2d(r.data(), b.data(), m * sizeof(T), st), "copy");
  CheckHip(prost_hip_memcpy_d2d(s.data(), x.data(), n * sizeof(T), st), "copy");
  double normx = Nrm2(x, n);
  if (normx > 0.) Gemv('n', (T)-1, x, (T)1, r);          // r = b - A x
  Gemv('t', (T)1, r, kNegShift, s);                      // s = A' r - shift x
  CheckHip(prost_hip_memcpy_d2d(p.data(), s.data(), n * sizeof(T), st), "copy");
  double norms = Nrm2(s, n);
  const double norms0 = norms;
  double gamma = norms0 * norms0;
  normx = Nrm2(x, n);
  double xmax = normx;
  if (norms < kEps) flag = 1;
  for (k = 0; k < maxit && !flag; ++k) {
    Gemv('n', (T)1, p, (T)0, q);                         // q = A p
    const double normp = Nrm2(p, n), normq = Nrm2(q, m);
    double dlt = normq * normq + shift * normp * normp;
    if (dlt <= 0.) indefinite = 1;
    if (dlt == 0.) dlt = kEps;
    const T alpha = (T)(gamma / dlt), neg_alpha = (T)(-gamma / dlt);
    CheckHip(Api<T>::axpy(x.data(), p.data(), (double)alpha, n, st), "axpy");        // x += alpha p
    CheckHip(Api<T>::axpy(r.data(), q.data(), (double)neg_alpha, m, st), "axpy");    // r -= alpha q
    CheckHip(prost_hip_memcpy_d2d(s.data(), x.data(), n * sizeof(T), st), "copy");
    Gemv('t', (T)1, r, kNegShift, s);
    norms = Nrm2(s, n);
    const double gamma1 = gamma;
    gamma = norms * norms;
    const T beta = (T)(gamma / gamma1);
    CheckHip(Api<T>::axpy(s.data(), p.data(), (double)beta, n, st), "axpy");         // p = s + beta p
    CheckHip(prost_hip_memcpy_d2d(p.data(), s.data(), n * sizeof(T), st), "copy");
    normx = Nrm2(x, n);
    xmax = std::max(xmax, normx);
    if ((norms <= norms0 * tol) || (normx * tol >= 1.)) break;
  }
  const double shrink = normx / xmax;
  if (k == maxit) flag = 2;
  else if (indefinite) flag = 3;
  else if (shrink * shrink <= tol) flag = 4;
  iterations = k;
  return flag;
}


/// The operator as a table of CSR / gradient blocks for prost_hip_cgls_round_*: every block must describe itself as one of
/// those (plugin blocks, diags, Kronecker blocks, label_first gradients and dualized operators do not), CSR blocks must be the
/// short-row kind the one-thread-per-row product is meant for (what prost_hip_csr_spmv itself picks up to 6 entries per row).
template <typename T>
void BackendADMM<T>::DescribeOperator() {
  auto linop = this->problem_->linop();
  if (dynamic_cast<DualLinearOperator<T>*>(linop.get())) return;
  const auto& blocks = linop->blocks();
  if (blocks.empty() || blocks.size() > (size_t)PROST_HIP_OP_MAX_BLOCKS) return;
  prost_hip_fused_op op;
  op.nblocks = 0;
  for (const auto& b : blocks) {
    BlockDesc bd;
    if (!b->describe(bd)) return;
    prost_hip_op_block& o = op.block[op.nblocks++];
    o.row = b->row(); o.col = b->col(); o.nrows = b->nrows(); o.ncols = b->ncols();
    o.nx = o.ny = o.L = 0;
    o.val = o.val_t = nullptr; o.ptr = o.ind = o.ptr_t = o.ind_t = nullptr;
    o.ids = o.ids_t = nullptr; o.pptr = o.rel = o.pptr_t = o.rel_t = nullptr; o.pval = o.pval_t = nullptr; o.anchor = o.anchor_t = nullptr;
    if (bd.kind == BlockDesc::kSparse) {
      if ((double)bd.nnz > 6.0 * (double)b->nrows() || (double)bd.nnz > 6.0 * (double)b->ncols()) return;
      o.kind = PROST_OP_CSR;
      o.val = bd.val; o.ptr = bd.ptr; o.ind = bd.ind; o.val_t = bd.val_t; o.ptr_t = bd.ptr_t; o.ind_t = bd.ind_t;
      o.ids = bd.ids; o.pptr = bd.pptr; o.rel = bd.rel; o.pval = bd.pval; o.ids_t = bd.ids_t; o.pptr_t = bd.pptr_t; o.rel_t = bd.rel_t; o.pval_t = bd.pval_t; o.anchor = bd.anchor; o.anchor_t = bd.anchor_t;
    } else if ((bd.kind == BlockDesc::kGradient2D || bd.kind == BlockDesc::kGradient3D) && !bd.label_first) {
      o.kind = bd.kind == BlockDesc::kGradient2D ? PROST_OP_GRAD2D : PROST_OP_GRAD3D;
      o.nx = bd.nx; o.ny = bd.ny; o.L = bd.L;
    } else {
      return;
    }
  }
  if (prost_hip_fused_op_supported(&op, this->problem_->nrows(), this->problem_->ncols()) != 1) return;
  fused_op_ = op;
  fused_rounds_ = true;
  // K = [D ; gradient2d(nx, ny, L)] (either order) with D coupling the L channels of one pixel, or the gradient alone: the CG rounds
  // of two launches (prost_hip_cgls_pixel_round_*)
  if (!opts_.pixel_rounds || blocks.size() > 2) return;
  const prost_hip_op_block* grad = nullptr; const prost_hip_op_block* dblk = nullptr;
  size_t planes = 0;
  for (size_t i = 0; i < blocks.size(); i++) {
    BlockDesc bd;
    blocks[i]->describe(bd);
    if (op.block[i].kind == PROST_OP_GRAD2D && !grad) grad = &op.block[i];
    else if (op.block[i].kind == PROST_OP_CSR && bd.pointwise_planes > 0 && bd.val && !dblk) { dblk = &op.block[i]; planes = bd.pointwise_planes; }
    // round 6: any CSR block with one row per pixel (a warp matrix that gathers at displaced pixels) -- both directions as plain CSR arrays
    else if (op.block[i].kind == PROST_OP_CSR && bd.val && bd.ptr && bd.ind && bd.val_t && bd.ptr_t && bd.ind_t && !dblk) { dblk = &op.block[i]; planes = 0; }
    else return;
  }
  if (!grad || grad->col != 0 || grad->L < 1 || grad->L > 3) return;
  prost_hip_pixel_op po;
  std::memset(&po, 0, sizeof(po));
  po.nx = grad->nx; po.ny = grad->ny; po.L = (int)grad->L; po.has_d = dblk ? 1 : 0;
  po.d_first = dblk && dblk < grad ? 1 : 0;                 // position in the block LIST (the order K^T t is accumulated in)
  po.g_row = grad->row; po.d_row = dblk ? dblk->row : 0; po.w = dblk ? dblk->val : nullptr;
  po.p_alt = po.r_alt = nullptr;
  if (dblk && (dblk->col != 0 || dblk->nrows != grad->nx * grad->ny)) return;
  if (dblk && planes == 0) {
    if (dblk->ncols != grad->L * grad->nx * grad->ny) return;
    po.d_csr = 1; po.w = nullptr;
    po.d_val = dblk->val; po.d_ptr = dblk->ptr; po.d_ind = dblk->ind; po.dt_val = dblk->val_t; po.dt_ptr = dblk->ptr_t; po.dt_ind = dblk->ind_t;
  } else if (dblk && planes != grad->L) {
    return;
  }
  // Sigma must be ONE value on the gradient rows (it is for the alpha-preconditioners: every row of a gradient block sums to 2,
  // block_gradient2d.cu:154-158; user-supplied scaling vectors may differ): the rounds read it as a scalar
  {
    const std::vector<T>& sl = this->problem_->scaling_left_host();
    const size_t g0 = (size_t)grad->row, g1 = g0 + (size_t)grad->nrows;
    if (sl.size() < g1) return;
    const T v0 = sl[g0];
    std::atomic<bool> same(true);
    ParallelFor(g1 - g0, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) if (sl[g0 + i] != v0) { same.store(false); return; } });
    if (!same.load()) return;
    po.sigma_grad = (double)v0;
  }
  if (prost_hip_pixel_op_supported(&po, this->problem_->nrows(), this->problem_->ncols(), sizeof(T) == 4 ? 0 : 1) != 1) return;
  cg_p_alt_.resize(this->problem_->ncols());
  cg_r_alt_.resize(this->problem_->nrows());
  po.p_alt = cg_p_alt_.data(); po.r_alt = cg_r_alt_.data();
  pixel_op_ = po;
  pixel_rounds_ = true;
}

/// The same solve with the CG scalars resident on the device (prost_hip_cgls_stage_*): no host round
/// trip per iteration.  All maxit rounds are queued unless the pinned stop word shows that the device
/// already met the stopping test; rounds queued after that point return immediately on the device.
template <typename T>
void BackendADMM<T>::CglsDevice(const device_vector<T>& b, device_vector<T>& x, double shift, double tol, int maxit,
                                device_vector<T>& p, device_vector<T>& q, device_vector<T>& r, device_vector<T>& s) {
  void* st = CurrentStream();
  prost_hip_cgls_desc d;
  d.state = cg_state_; d.workspace = cg_workspace_;
  d.b = b.data(); d.x = x.data(); d.p = p.data(); d.q = q.data(); d.r = r.data(); d.s = s.data(); d.t = temp3_.data();
  d.sigma = this->problem_->scaling_left().data(); d.tau = this->problem_->scaling_right().data();
  d.m = this->problem_->nrows(); d.n = this->problem_->ncols();
  d.shift = shift; d.tol = tol;
  d.host_done = cg_done_host_; d.epoch = ++cg_epoch_;
  auto stage = [&](int which) { CheckHip(Api<T>::cgls_stage(which, &d, st), "cgls_stage"); };
  LinearOperator<T>* K = this->problem_->linop().get();
  auto round = [&]() {
    K->Eval(q, temp3_, 0);
    stage(PROST_CGLS_STEP_Q);
    stage(PROST_CGLS_STEP_XR);
    K->EvalAdjoint(s, temp3_, 1);
    stage(PROST_CGLS_STEP_S);
    stage(PROST_CGLS_STEP_P);
  };
  cg_result_index_ = 0;
  if (fused_rounds_) {
    // the operator inside the kernels: three launches + two scalar kernels to start, four launches per round; record j + 1
    // is written by round j
    CheckHip(Api<T>::cgls_init_fused(&d, &fused_op_, st), "cgls_init_fused");
    int queued = 0;
    // kernel timing: the second round of one solve in `sample_every_` is bracketed kernel by kernel (at most 512 samples)
    const bool sample = this->time_kernels_ && (solves_++ % (size_t)this->sample_every_) == 0 && ev_used_ + 8 <= 8 * 512;
    for (int k = 0; k < maxit; ++k) {
      if (*static_cast<volatile int*>(cg_done_host_) == d.epoch) break;
      if (sample && k == (maxit > 1 ? 1 : 0)) {
        while (ev_.size() < ev_used_ + 8) { void* e; CheckHip(prost_hip_event_create(&e), "event_create"); ev_.push_back(e); }
        if (pixel_rounds_) {
          // two launches per round: events 0-3 of the group of eight are used, 4-7 stay unrecorded (KernelTimes reads two kernels)
          CheckHip(Api<T>::cgls_pixel_round_timed(&d, &pixel_op_, k, ev_.data() + ev_used_, st), "cgls_pixel_round");
        } else {
          CheckHip(Api<T>::cgls_round_timed(&d, &fused_op_, k, ev_.data() + ev_used_, st), "cgls_round");
        }
        ev_used_ += 8;
      } else if (pixel_rounds_) {
        CheckHip(Api<T>::cgls_pixel_round(&d, &pixel_op_, k, st), "cgls_pixel_round");
      } else {
        CheckHip(Api<T>::cgls_round(&d, &fused_op_, k, st), "cgls_round");
      }
      queued++;
    }
    // the last queued round's beta / stopping test -> record `queued` (the four-launch rounds write it themselves)
    if (pixel_rounds_ && queued > 0) CheckHip(Api<T>::cgls_pixel_close(&d, &pixel_op_, queued - 1, st), "cgls_pixel_close");
    if (this->time_kernels_) rounds_launched_ += (size_t)queued;
    cg_result_index_ = queued;
    cg_iters_valid_ = false;
    return;
  }
  stage(PROST_CGLS_INIT_X);
  stage(PROST_CGLS_INIT_R);
  K->Eval(r, temp3_, 1);
  stage(PROST_CGLS_INIT_R2);
  K->EvalAdjoint(s, temp3_, 1);
  stage(PROST_CGLS_INIT_S);
  if (opts_.cg_graph && maxit > 0) {
    // The maxit rounds take no per-solve argument (tolerance and epoch sit in the device record, the vectors are
    // members), so they can be captured ONCE into a HIP graph and replayed with one host call per solve.
    // Opt-in: on ROCm 7.2 / MI355X the replay of the 80-node graph measured 0.11 ms slower per solve than the direct
    // launches at every size tried (256^2: 0.57 vs 0.45 ms per outer iteration), see DESIGN.md.
    if (!cg_graph_) {
      void* prev = CurrentStream();
      SetCurrentStream(cg_stream_);
      st = cg_stream_;
      try {
        CheckHip(prost_hip_stream_begin_capture(cg_stream_), "begin_capture");
        for (int k = 0; k < maxit; ++k) round();
        CheckHip(prost_hip_stream_end_capture(cg_stream_, &cg_graph_), "end_capture");
      } catch (...) { SetCurrentStream(prev); throw; }
      SetCurrentStream(prev);
      st = prev;
    }
    CheckHip(prost_hip_event_record(cg_ev_[0], st), "event_record");
    CheckHip(prost_hip_stream_wait_event(cg_stream_, cg_ev_[0]), "stream_wait_event");
    CheckHip(prost_hip_graph_launch(cg_graph_, cg_stream_), "graph_launch");
    CheckHip(prost_hip_event_record(cg_ev_[1], cg_stream_), "event_record");
    CheckHip(prost_hip_stream_wait_event(st, cg_ev_[1]), "stream_wait_event");
  } else {
    for (int k = 0; k < maxit; ++k) {
      if (*static_cast<volatile int*>(cg_done_host_) == d.epoch) break;
      round();
    }
  }
  cg_iters_valid_ = false;
}

/// mean duration of the four kernels of a CG round over the rounds sampled since the last call (rounds that ran after the
/// stopping test fired return at once and would show up as ~2 us launches: the sampled round is the second of a solve)
template <typename T>
void BackendADMM<T>::KernelTimes(std::vector<typename Backend<T>::KernelTime>& out) {
  out.clear();
  if (ev_used_ == 0) return;
  CheckHip(prost_hip_event_synchronize(ev_[pixel_rounds_ ? ev_used_ - 5 : ev_used_ - 1]), "event_synchronize");      // the last RECORDED event of the last group
  static const char* const names4[4] = {"op_stage_kernel<EpiFwdQ>", "cg_step_xr2_kernel", "op_stage_kernel<EpiAdjS>", "cg_step_p2_kernel"};
  static const char* const names2[4] = {"cg_pixel_pq_kernel", "cg_pixel_xrs_kernel", "", ""};
  const char* const* names = pixel_rounds_ ? names2 : names4;
  const int kernels = pixel_rounds_ ? 2 : 4;
  double sum[4] = {0, 0, 0, 0};
  const size_t samples = ev_used_ / 8;
  for (size_t s = 0; s < samples; s++)
    for (int k = 0; k < kernels; k++) {                       // the kernel's own begin / end stamps (hipExtLaunchKernel): no marker in between
      float ms = 0;
      CheckHip(prost_hip_event_elapsed_ms(ev_[8 * s + 2 * k], ev_[8 * s + 2 * k + 1], &ms), "event_elapsed");
      sum[k] += ms;
    }
  for (int k = 0; k < kernels; k++) out.push_back({names[k], sum[k] / (double)samples, samples, rounds_launched_, 0, 0});
  ev_used_ = 0; rounds_launched_ = 0; solves_ = 0;
}

template <typename T>
int BackendADMM<T>::last_cg_iterations() {
  if (!cg_iters_valid_ && cg_state_) {
    prost_hip_cgls_result_t res;
    CheckHip(prost_hip_cgls_result_at(cg_state_, cg_result_index_, &res, CurrentStream()), "cgls_result");
    last_cg_iters_ = res.iterations;
    cg_iters_valid_ = true;
  }
  return last_cg_iters_;
}

template <typename T>
void BackendADMM<T>::GetDual(device_vector<T>& out, const device_vector<T>& half, const device_vector<T>& proj,
                             const device_vector<T>& dual, const device_vector<T>& scaling, T expo, size_t n) {
  // out = -rho * scaling^expo * (half - proj + dual)   (get_dual_functor, backend_admm.cu:181-196)
  elem<T>(PROST_ADMM_GETDUAL, out.data(), half.data(), proj.data(), dual.data(), scaling.data(), (double)rho_, (double)expo, n);
}

template <typename T>
void BackendADMM<T>::PerformIteration() {
  if (opts_.device_cg) PerformIterationFused(); else PerformIterationUnfused();
}

/// residual bookkeeping shared by both paths: all-reduce over ranks, rho adaptation (:618-663)
template <typename T>
void BackendADMM<T>::FinishResiduals(double primal_residual, double primal_var_norm, double dual_residual, double dual_var_norm) {
  const size_t m = this->problem_->nrows(), n = this->problem_->ncols();
  void* st = CurrentStream();
  if (this->comm_) {
    scal_host_[0] = primal_residual * primal_residual; scal_host_[1] = primal_var_norm * primal_var_norm;
    scal_host_[2] = dual_residual * dual_residual; scal_host_[3] = dual_var_norm * dual_var_norm;
    CheckHip(prost_hip_memcpy_h2d(scal_dev_, scal_host_, 4 * sizeof(double), st), "h2d");
    CheckHip(prost_hip_allreduce_sum_f64(this->comm_, scal_dev_, 4, st), "allreduce");
    CheckHip(prost_hip_memcpy_d2h(scal_host_, scal_dev_, 4 * sizeof(double), st), "d2h");
    CheckHip(prost_hip_stream_synchronize(st), "sync");
    primal_residual = std::sqrt(scal_host_[0]); primal_var_norm = std::sqrt(scal_host_[1]);
    dual_residual = std::sqrt(scal_host_[2]); dual_var_norm = std::sqrt(scal_host_[3]);
  }
  this->primal_residual_ = (T)primal_residual;
  this->primal_var_norm_ = (T)primal_var_norm;
  this->dual_residual_ = (T)dual_residual;
  this->dual_var_norm_ = (T)dual_var_norm;

  const T eps_primal = this->eps_primal(), eps_dual = this->eps_dual();
  const T rho_prev = rho_;
  if ((this->dual_residual_ < eps_dual) && (opts_.arb_tau * iteration_ > arb_l_)) {
    rho_ *= delta_; delta_ *= opts_.arb_gamma; arb_u_ = (int)iteration_;
  } else if ((this->primal_residual_ < eps_primal) && (opts_.arb_tau * iteration_ > arb_u_)) {
    rho_ /= delta_; delta_ *= opts_.arb_gamma; arb_l_ = (int)iteration_;
  }
  if (std::abs(rho_ - rho_prev) > 1e-7) {                                                            // :650-663
    const T f = rho_prev / rho_;
    elem<T>(PROST_ADMM_SCALE, x_dual_.data(), x_dual_.data(), nullptr, nullptr, nullptr, (double)f, 0, n);
    elem<T>(PROST_ADMM_SCALE, z_dual_.data(), z_dual_.data(), nullptr, nullptr, nullptr, (double)f, 0, m);
  }
  CheckHip(prost_hip_check_last_error(), "ADMM iteration");
}

/// The iteration on the fused passes of prost_hip_admm_stage_* with the device-resident CGLS: the same
/// per-element expressions as PerformIterationUnfused, 7 passes + 2 prox + 5 operator applications
/// outside the CG solve instead of ~45 launches, and ONE host synchronisation (the four residual norms).
template <typename T>
void BackendADMM<T>::PerformIterationFused() {
  const size_t m = this->problem_->nrows(), n = this->problem_->ncols();
  const device_vector<T>& Sl = this->problem_->scaling_left();
  const device_vector<T>& Tr = this->problem_->scaling_right();
  void* st = CurrentStream();
  LinearOperator<T>* K = this->problem_->linop().get();
  prost_hip_admm_desc d;
  d.workspace = cg_workspace_;
  d.x_half = x_half_.data(); d.x_proj = x_proj_.data(); d.x_dual = x_dual_.data();
  d.z_half = z_half_.data(); d.z_proj = z_proj_.data(); d.z_dual = z_dual_.data();
  d.temp1 = temp1_.data(); d.temp2 = temp2_.data(); d.temp3 = temp3_.data();
  d.kx = tmp_m_.data(); d.kty = tmp_n_.data();
  d.sigma = Sl.data(); d.tau = Tr.data();
  d.m = m; d.n = n;
  d.alpha = (double)(T)opts_.alpha; d.rho = (double)rho_;
  d.out4 = scal_host_;
  auto stage = [&](int which) { CheckHip(Api<T>::admm_stage(which, &d, st), "admm_stage"); };

  auto fused = [&](int which) { CheckHip(Api<T>::admm_fused_stage(which, &d, &fused_op_, st), "admm_fused_stage"); };
  if (fused_rounds_) fused(PROST_ADMM_FUSED_PRE);
  else {
    stage(PROST_ADMM_STAGE_PRE_X);
    stage(PROST_ADMM_STAGE_PRE_Z);
    K->Eval(z_dual_, temp3_, 1);
    stage(PROST_ADMM_STAGE_PRE_Z2);
  }

  double cg_tol = opts_.cg_tol_min / std::pow((double)static_cast<T>(iteration_ + 1), opts_.cg_tol_pow);   // :408-410
  cg_tol = std::max(cg_tol, opts_.cg_tol_max);
  CglsDevice(z_dual_, x_proj_, 1, cg_tol, opts_.cg_max_iter, x_half_, z_half_, z_proj_, x_dual_);

  if (fused_rounds_) fused(PROST_ADMM_FUSED_POST);
  else {
    stage(PROST_ADMM_STAGE_POST_X);
    K->Eval(z_proj_, x_proj_);
    stage(PROST_ADMM_STAGE_POST_XZ);
  }
  // an identity prox over the whole variable (prox_zero.cu:37-48: a device copy of the argument) is a buffer exchange here:
  // temp1 is rewritten from scratch by the next PRE stage (not under a captured graph, which holds the buffer addresses)
  if (!opts_.cg_graph && prox_g_.size() == 1 && dynamic_cast<ProxZero<T>*>(prox_g_[0].get()) && prox_g_[0]->index() == 0 && prox_g_[0]->size() == n)
    x_half_.swap(temp1_);
  else
    for (auto& p : prox_g_) p->Eval(x_half_, temp1_, Tr, 1 / rho_);
  for (auto& p : prox_f_) p->Eval(z_half_, temp2_, Sl, rho_, true);

  iteration_++;

  if (iteration_ == 0 || (iteration_ % (size_t)opts_.residual_iter) == 0) {                            // :535-616
    d.x_half = x_half_.data(); d.temp1 = temp1_.data();
    if (fused_rounds_) fused(PROST_ADMM_FUSED_RES);
    else {
      K->Eval(tmp_m_, x_half_);
      stage(PROST_ADMM_STAGE_RES_Z);
      K->EvalAdjoint(tmp_n_, tmp_m_);
      stage(PROST_ADMM_STAGE_RES_X);                     // the fold writes the four norms to pinned host memory
    }
    CheckHip(prost_hip_stream_synchronize(st), "sync");
    FinishResiduals((double)(T)scal_host_[0], (double)(T)scal_host_[1], (double)(T)scal_host_[2], (double)(T)scal_host_[3]);
  }
}

/// The reference's own sequence, one launch per functor and one blocking copy per norm
/// (kept selectable -- backend option device_cg = false -- as the A/B for the fused path).
template <typename T>
void BackendADMM<T>::PerformIterationUnfused() {
  const size_t m = this->problem_->nrows(), n = this->problem_->ncols();
  const device_vector<T>& Sl = this->problem_->scaling_left();
  const device_vector<T>& Tr = this->problem_->scaling_right();
  void* st = CurrentStream();

  elem<T>(PROST_ADMM_TEMP1, temp1_.data(), x_half_.data(), x_proj_.data(), x_dual_.data(), Tr.data(), (double)(T)opts_.alpha, 0, n);   // :358-373
  elem<T>(PROST_ADMM_TEMP2, temp2_.data(), z_half_.data(), z_dual_.data(), Sl.data(), nullptr, 0, 0, m);                               // :376-389
  CheckHip(prost_hip_memcpy_d2d(z_dual_.data(), temp2_.data(), m * sizeof(T), st), "copy");          // projection argument lives in z_dual_
  CheckHip(prost_hip_memcpy_d2d(x_proj_.data(), temp3_.data(), n * sizeof(T), st), "copy");          // CG warm start
  Gemv('n', (T)-1, temp1_, (T)1, z_dual_);

  double cg_tol = opts_.cg_tol_min / std::pow((double)static_cast<T>(iteration_ + 1), opts_.cg_tol_pow);   // :408-410
  cg_tol = std::max(cg_tol, opts_.cg_tol_max);
  Cgls(z_dual_, x_proj_, 1, cg_tol, opts_.cg_max_iter, x_half_, z_half_, z_proj_, x_dual_, last_cg_iters_);
  cg_iters_valid_ = true;

  CheckHip(prost_hip_memcpy_d2d(temp3_.data(), x_proj_.data(), n * sizeof(T), st), "copy");
  elem<T>(PROST_ADMM_XPROJ, x_proj_.data(), temp1_.data(), Tr.data(), nullptr, nullptr, 0, 0, n);                                       // :447-456
  this->problem_->linop()->Eval(z_proj_, x_proj_);
  elem<T>(PROST_ADMM_XDUAL, x_dual_.data(), temp1_.data(), x_proj_.data(), Tr.data(), nullptr, 0, 0, n);                                // :464-477
  elem<T>(PROST_ADMM_ZDUAL, z_dual_.data(), temp2_.data(), z_proj_.data(), Sl.data(), nullptr, 0, 0, m);                                // :480-493
  elem<T>(PROST_ADMM_DIFF, temp1_.data(), x_proj_.data(), x_dual_.data(), nullptr, nullptr, 0, 0, n);
  for (auto& p : prox_g_) p->Eval(x_half_, temp1_, Tr, 1 / rho_);
  elem<T>(PROST_ADMM_DIFF, temp2_.data(), z_proj_.data(), z_dual_.data(), nullptr, nullptr, 0, 0, m);
  for (auto& p : prox_f_) p->Eval(z_half_, temp2_, Sl, rho_, true);

  iteration_++;

  if (iteration_ == 0 || (iteration_ % (size_t)opts_.residual_iter) == 0) {                            // :535-663
    CheckHip(prost_hip_memcpy_d2d(temp2_.data(), z_half_.data(), m * sizeof(T), st), "copy");
    {   // temp2 = z_half - K x_half  (Eval with beta = -1 on the m-prefix of temp2_)
      elem<T>(PROST_ADMM_SCALE, temp2_.data(), temp2_.data(), nullptr, nullptr, nullptr, -1.0, 0, m);
      this->problem_->linop()->Eval(tmp_m_, x_half_);
      CheckHip(Api<T>::axpy(temp2_.data(), tmp_m_.data(), 1.0, m, st), "axpy");
    }
    elem<T>(PROST_ADMM_GEMV1, temp2_.data(), Sl.data(), temp2_.data(), nullptr, nullptr, 0, 0, m);
    double primal_residual = (double)(T)Nrm2(temp2_, m);
    elem<T>(PROST_ADMM_GEMV1, temp2_.data(), Sl.data(), z_half_.data(), nullptr, nullptr, 0, 0, m);
    double primal_var_norm = (double)(T)Nrm2(temp2_, m);
    GetDual(temp1_, x_half_, x_proj_, x_dual_, Tr, (T)-1, n);                                          // w
    elem<T>(PROST_ADMM_GEMV1, temp2_.data(), Tr.data(), temp1_.data(), nullptr, nullptr, 0, 0, n);
    double dual_var_norm = (double)(T)Nrm2(temp2_, n);
    GetDual(temp2_, z_half_, z_proj_, z_dual_, Sl, (T)1, m);                                           // y
    {   // temp1 = w + K^T y
      CheckHip(prost_hip_memcpy_d2d(tmp_m_.data(), temp2_.data(), m * sizeof(T), st), "copy");
      this->problem_->linop()->EvalAdjoint(tmp_n_, tmp_m_);
      CheckHip(Api<T>::axpy(temp1_.data(), tmp_n_.data(), 1.0, n, st), "axpy");
    }
    elem<T>(PROST_ADMM_GEMV1, temp1_.data(), Tr.data(), temp1_.data(), nullptr, nullptr, 0, 0, n);
    double dual_residual = (double)(T)Nrm2(temp1_, n);

    FinishResiduals(primal_residual, primal_var_norm, dual_residual, dual_var_norm);
  }
}

template <typename T>
void BackendADMM<T>::current_solution(std::vector<T>& primal, std::vector<T>& dual) {
  const size_t m = this->problem_->nrows();
  x_half_.copy_to(primal);
  GetDual(temp2_, z_half_, z_proj_, z_dual_, this->problem_->scaling_left(), (T)1, m);
  std::vector<T> tmp; temp2_.copy_to(tmp);
  dual.assign(tmp.begin(), tmp.begin() + m);
}

template <typename T>
void BackendADMM<T>::current_solution(std::vector<T>& primal_x, std::vector<T>& primal_z, std::vector<T>& dual_y, std::vector<T>& dual_w) {
  const size_t m = this->problem_->nrows(), n = this->problem_->ncols();
  GetDual(temp1_, x_half_, x_proj_, x_dual_, this->problem_->scaling_right(), (T)-1, n);
  temp1_.copy_to(dual_w);
  GetDual(temp2_, z_half_, z_proj_, z_dual_, this->problem_->scaling_left(), (T)1, m);
  std::vector<T> tmp; temp2_.copy_to(tmp);
  dual_y.assign(tmp.begin(), tmp.begin() + m);
  x_half_.copy_to(primal_x);
  z_half_.copy_to(primal_z);
}

template <typename T>
size_t BackendADMM<T>::gpu_mem_amount() const {
  const size_t m = this->problem_->nrows(), n = this->problem_->ncols();
  return (4 * (n + m) + std::max(m, n)) * sizeof(T) + (cg_p_alt_.size() + cg_r_alt_.size()) * sizeof(T);      // (+ the second p / r buffers of the two-launch rounds)
}

template class BackendADMM<float>;
template class BackendADMM<double>;

}  // namespace prost
