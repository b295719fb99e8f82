// backend_pdhg.cpp -- preconditioned PDHG on the MI355X kernels.
//
// Iteration structure and every scalar update follow the reference (src/backend/backend_pdhg.cu);
// what differs is where the vector work happens:
//   fused path   : prost_hip_fused_primal + prost_hip_fused_dual, state = {x, x_prev, y, y_prev}
//   generic path : the reference's sequence (prox arg, prox, K, dual arg, prox, K^T) on the
//                  generic kernels, state = the reference's nine vectors
// Residual sums are produced on the device into res_dev_[0..3]; on residual iterations they are
// (optionally all-reduced over RCCL and) copied to pinned host memory, which is the only host
// synchronisation of the loop (the reference synchronises after every prox and twice per
// residual iteration).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <atomic>
#include <cmath>
#include <cstring>
#include <iostream>

#include "hipapi.hpp"
#include "prost/backend/backend_pdhg.hpp"
#include "prost/prox/proxes.hpp"

namespace prost {

template <typename T>
BackendPDHG<T>::~BackendPDHG() { Release(); }

template <typename T>
std::string BackendPDHG<T>::path() const {
  if (!fused_) return "pdhg:generic";
  if (from_matrix_) return "pdhg:fused-grad2d(sparse)";        // the stencil kernels on a gradient handed over as a sparse matrix
  if (arithmetic() == PROST_HIP_ARITH_FMAD) return desc_.is3d ? "pdhg:fused-grad3d+fmad" : "pdhg:fused-grad2d+fmad";      // ... in the pair launches
  return desc_.is3d ? "pdhg:fused-grad3d" : "pdhg:fused-grad2d";
}

template <typename T>
static bool uniform(const std::vector<T>& v, T& value) {
  if (v.empty()) return false;
  value = v[0];
  const T first = value;
  std::atomic<bool> same(true);
  ParallelFor(v.size(), [&](size_t b, size_t e) {
    for (size_t i = b; i < e && same.load(std::memory_order_relaxed); i++) if (v[i] != first) { same.store(false); return; }
  });
  return same.load();
}

/// prox_g as ONE elem_operation:1d over all primal entries.  The primal variable may carry several of them on consecutive
/// sub-variables (example_rof_primal.m:19-26: u1, u2, u3 with slices of f as coefficient b): same function, each coefficient a scalar
/// or a vector per piece.  A coefficient that is the same scalar in every piece stays a scalar; any other becomes ONE vector over the
/// whole variable, assembled on the device from the pieces' vectors / scalars (merged_g_) -- what the kernels stream either way.
template <typename T>
bool BackendPDHG<T>::DescribeProxG(ProxDesc& out) {
  const size_t n = this->problem_->ncols();
  for (auto& v : merged_g_) v.clear();
  if (prox_g_.size() == 1) return prox_g_[0]->describe(out) && prox_g_[0]->index() == 0 && prox_g_[0]->size() == n;
  std::vector<std::pair<size_t, ProxDesc>> pieces;             // (index, description), sorted by index
  for (auto& p : prox_g_) {
    ProxDesc d;
    if (!p->describe(d) || d.kind != ProxDesc::kElem1D || d.moreau || d.count != p->size()) return false;
    pieces.push_back({p->index(), d});
  }
  std::sort(pieces.begin(), pieces.end(), [](const std::pair<size_t, ProxDesc>& a, const std::pair<size_t, ProxDesc>& b) { return a.first < b.first; });
  size_t at = 0;
  for (auto& pc : pieces) {
    if (pc.first != at || pc.second.fn != pieces[0].second.fn) return false;
    at += pc.second.count;
  }
  if (at != n) return false;
  out = pieces[0].second;
  out.count = n;
  for (int k = 0; k < 7; k++) {
    bool scalar = true;
    for (auto& pc : pieces) scalar = scalar && !pc.second.coeff_ptr[k] && pc.second.coeff_val[k] == pieces[0].second.coeff_val[k];
    if (scalar) { out.coeff_ptr[k] = nullptr; out.coeff_val[k] = pieces[0].second.coeff_val[k]; continue; }
    merged_g_[k].resize(n);
    for (auto& pc : pieces) {
      T* dst = merged_g_[k].data() + pc.first;
      if (pc.second.coeff_ptr[k]) CheckHip(prost_hip_memcpy_d2d(dst, pc.second.coeff_ptr[k], pc.second.count * sizeof(T), CurrentStream()), "memcpy_d2d");
      else CheckHip(Api<T>::fill(dst, pc.second.coeff_val[k], pc.second.count, CurrentStream()), "fill");
    }
    out.coeff_ptr[k] = merged_g_[k].data(); out.coeff_val[k] = 0;
  }
  return true;
}

/// fused path applies iff: one gradient2d/3d block (not label_first) spanning the whole operator,
/// one elem_operation:1d prox_g over all primal entries, one elem_operation:norm2 prox_f* whose
/// groups are the planar gradient components of a pixel, uniform Sigma and Tau.
template <typename T>
bool BackendPDHG<T>::TryFused() {
  if (!opts_.allow_fused) return false;
  auto& prob = *this->problem_;
  auto linop = prob.linop();
  if (linop->blocks().size() != 1 || prox_g_.empty() || prox_fstar_.size() != 1) return false;
  BlockDesc bd;
  auto blk = linop->blocks()[0];
  // a block whose OPERATOR is gradient2d although it was handed over as a sparse matrix (example_rof_primal.m:10, :28): the stencil
  // kernels apply, with the preconditioners that matrix really has (checked against the problem's vectors below)
  bool as_matrix = false;
  from_matrix_ = false;
  if (blk->describe(bd) && (bd.kind == BlockDesc::kGradient2D || bd.kind == BlockDesc::kGradient3D)) {}
  else if (blk->stencil_shape(bd) && bd.kind == BlockDesc::kGradient2D && bd.L >= 1 && bd.L <= 4) as_matrix = true;
  else return false;
  if (bd.label_first) return false;
  if (blk->row() != 0 || blk->col() != 0 || blk->nrows() != prob.nrows() || blk->ncols() != prob.ncols()) return false;
  ProxDesc pg, pf;
  if (!DescribeProxG(pg) || !prox_fstar_[0]->describe(pf)) return false;
  // prox_f* may be the Moreau wrap of prox_f (a problem written in the primal form, example_rof_primal.m:27): one-kernel iterations
  if (pg.kind != ProxDesc::kElem1D || pg.moreau || pf.kind != ProxDesc::kElemNorm2 || pf.interleaved) return false;
  if (prox_fstar_[0]->index() != 0 || prox_fstar_[0]->size() != prob.nrows()) return false;
  const bool d3 = bd.kind == BlockDesc::kGradient3D;
  const size_t pixels = d3 ? bd.nx * bd.ny * bd.L : bd.nx * bd.ny;
  const size_t comps = d3 ? 3 : 2 * bd.L;
  if (pf.count != pixels || pf.dim != comps) return false;
  T tv, sv;
  desc_.var_T = 0; desc_.T_cls[0] = desc_.T_cls[1] = desc_.T_cls[2] = 0;
  if (!prob.uniform_left(sv) && !uniform(prob.scaling_left_host(), sv)) return false;
  if (as_matrix) {
    // Tau_j = 1 / (column sum of |K|) (problem.cu:262-287): 4 stencil entries in the column of an interior pixel, 3 on an edge, 2 in a
    // corner.  The three values are READ from the problem's vector and every entry is compared with the value of its class.
    if (owned_x1_ != 0 || bd.nx < 4 || bd.ny < 4) return false;
    const std::vector<T>& tr = prob.scaling_right_host();
    const size_t nx = bd.nx, ny = bd.ny;
    if (tr.size() != nx * ny * bd.L) return false;
    const T cls[3] = {tr[0], tr[1], tr[ny + 1]};            // corner, edge, interior
    std::atomic<bool> same(true);
    ParallelFor(nx * bd.L, [&](size_t b, size_t e) {         // (every channel repeats the one-channel matrix)
      for (size_t xl = b; xl < e && same.load(std::memory_order_relaxed); xl++) {
        const size_t x = xl % nx;
        for (size_t y = 0; y < ny; y++) {
          const int cnt = 4 - (x == 0) - (x == nx - 1) - (y == 0) - (y == ny - 1);
          if (tr[xl * ny + y] != cls[cnt - 2]) { same.store(false); return; }
        }
      }
    });
    if (!same.load()) return false;
    tv = cls[2];
    desc_.var_T = (cls[0] != cls[2] || cls[1] != cls[2]) ? 1 : 0;
    for (int i = 0; i < 3; i++) desc_.T_cls[i] = (double)cls[i];
  }
  else if (!prob.uniform_right(tv) && !uniform(prob.scaling_right_host(), tv)) return false;
  desc_.res_x0 = owned_x0_; desc_.res_x1 = owned_x1_;
  desc_.g_b_masked = 0;
  desc_.arith = PROST_HIP_ARITH_EXACT;
  desc_.is3d = d3 ? 1 : 0; desc_.nx = bd.nx; desc_.ny = bd.ny; desc_.L = bd.L;
  desc_.g_fn = pg.fn; desc_.f_fn = pf.fn; desc_.f_moreau = pf.moreau ? 1 : 0;
  for (int i = 0; i < 7; i++) {
    desc_.g_coeff_ptr[i] = pg.coeff_ptr[i]; desc_.g_coeff_val[i] = pg.coeff_val[i];
    desc_.f_coeff_ptr[i] = pf.coeff_ptr[i]; desc_.f_coeff_val[i] = pf.coeff_val[i];
  }
  desc_.T_val = (double)tv; desc_.S_val = (double)sv;
  // (position-dependent Tau, Moreau-wrapped prox_f*: the one-kernel iteration is the only fused form; the two-pass kernels refuse them)
  from_matrix_ = as_matrix;
  if (desc_.var_T || desc_.f_moreau) return opts_.allow_single_kernel && (prost_hip_fused_iteration_supported(&desc_, dtype_id<T>()) == 1 ||
                                                        prost_hip_fused_iteration_mc_supported(&desc_, dtype_id<T>()) == 1);
  return prost_hip_fused_supported(&desc_, dtype_id<T>()) == 1;
}

/// example_tv_inpaint.m:23 -- sum_1d('square', m, f, lmb) with a 0 / 1 mask m as coefficient a: ElemOperation1D skips the function
/// where a == 0 (elem_operation_1d.hpp:42-44), so with d = e = 0 a masked pixel passes through and every other pixel is the ROF
/// shape.  If a is binary (checked on the device, once) it is folded into the b stream -- sentinel where a == 0 -- and the
/// double-iteration kernels run their straight-line instance on that stream; otherwise they are not used for this problem.
template <typename T>
void BackendPDHG<T>::TryMaskedPairShape() {
  desc_pair_ = desc_;
  if (!fused_ || desc_.is3d || !desc_.g_coeff_ptr[0] || desc_.g_fn != PROST_FN_SQUARE) return;
  for (int k = 2; k < 7; k++) if (desc_.g_coeff_ptr[k]) return;
  if (desc_.g_coeff_val[2] == 0.0 || desc_.g_coeff_val[3] != 0.0 || desc_.g_coeff_val[4] != 0.0) return;
  const size_t n = this->problem_->ncols();
  b_masked_.resize(n);
  unsigned long long* counter = nullptr;
  CheckHip(prost_hip_malloc((void**)&counter, sizeof(unsigned long long)), "malloc");
  CheckHip(prost_hip_memset(counter, 0, sizeof(unsigned long long), CurrentStream()), "memset");
  CheckHip(Api<T>::mask_merge(b_masked_.data(), static_cast<const T*>(desc_.g_coeff_ptr[0]), static_cast<const T*>(desc_.g_coeff_ptr[1]),
                              desc_.g_coeff_val[1], n, counter, CurrentStream()), "mask_merge");
  unsigned long long nonbinary = 1;
  CheckHip(prost_hip_memcpy_d2h(&nonbinary, counter, sizeof(nonbinary), CurrentStream()), "memcpy_d2h");
  CheckHip(prost_hip_stream_synchronize(CurrentStream()), "stream_synchronize");
  prost_hip_free(counter);
  if (nonbinary != 0) { b_masked_.clear(); return; }
  desc_pair_.g_coeff_ptr[0] = nullptr; desc_pair_.g_coeff_val[0] = 1.0;
  desc_pair_.g_coeff_ptr[1] = b_masked_.data();
  desc_pair_.g_b_masked = 1;
}

template <typename T>
void BackendPDHG<T>::Initialize() {
  const size_t m = this->problem_->nrows(), n = this->problem_->ncols(), l = std::max(m, n);

  iteration_ = 0;
  failed_ = false;
  pair_launches_ = 0;
  spec_launched_ = spec_adopted_ = 0;
  prev_stale_ = false;
  spec_valid_ = false;
  residuals_pending_ = false;
  tau_ = (T)opts_.tau0;
  sigma_ = (T)opts_.sigma0;
  theta_ = 1;
  arb_l_ = arb_u_ = 0;
  arg_alpha_ = opts_.arg_alpha0;

  // missing prox_g / prox_f* are derived from the conjugates by Moreau (backend_pdhg.cu:236-266)
  prox_g_.clear(); prox_fstar_.clear();
  if (this->problem_->prox_g().empty()) {
    if (this->problem_->prox_gstar().empty()) throw Exception("Neither prox_g nor prox_gstar specified.");
    for (auto& p : this->problem_->prox_gstar()) { auto mo = std::make_shared<ProxMoreau<T>>(p); mo->Initialize(); prox_g_.push_back(mo); }
  } else prox_g_ = this->problem_->prox_g();
  if (this->problem_->prox_fstar().empty()) {
    if (this->problem_->prox_f().empty()) throw Exception("Neither prox_f nor prox_fstar specified.");
    for (auto& p : this->problem_->prox_f()) { auto mo = std::make_shared<ProxMoreau<T>>(p); mo->Initialize(); prox_fstar_.push_back(mo); }
  } else prox_fstar_ = this->problem_->prox_fstar();

  fused_ = TryFused();
  if (!fused_) for (auto& v : merged_g_) v.clear();
  arg_fused_g_ = arg_fused_f_ = opts_.allow_arg_fusion;
  for (auto& p : prox_g_) arg_fused_g_ = arg_fused_g_ && p->supports_arg_source();
  for (auto& p : prox_fstar_) arg_fused_f_ = arg_fused_f_ && p->supports_arg_source();
  // the operator inside the prox kernels (round 5): every block a sparse matrix or a gradient stencil, every prox able to form K^T y /
  // K x for its own elements (the in-tree elem operations, their Moreau wraps, the identity, on 16-byte boundaries)
  // (off by default: with the row walk of round 5 in the stand-alone products as well, the separate products are as fast or faster --
  // deblurring's shape 256^2 25.0 k against 22.5 k iterations/s, 1024^2 14.4 k against 14.9 k, 2048^2 5.7 k (residual sums in the prox
  // launches) against 5.4 k; example_multilabel_fast.m 512^2 24.9 k against 21.2 k)
  op_fused_ = !fused_ && opts_.allow_op_fusion > 0 && arg_fused_g_ && arg_fused_f_ && owned_x1_ == 0 && DescribeGenericOperator(false);
  for (auto& p : prox_g_) op_fused_ = op_fused_ && p->supports_op_source();
  for (auto& p : prox_fstar_) op_fused_ = op_fused_ && p->supports_op_source();
  // the residual sums inside the prox launches of the separate-products path too (round 5, ARG 5 / 6 of the prox kernels): every prox forms
  // its argument on the fly with the 16-bytes-per-lane kernel -- the separate reduction re-read eight vectors per residual iteration
  // (from 2^23 elements of x and y on: 2048^2 deblurring shape 5 137 -> 5 668 iterations/s; at 256^2 the sums' tail in every prox launch and the
  // 1024-lane fold cost more than the two small reduction launches they replace: 25.0 k -> 21.0 k)
  res_in_prox_ = !fused_ && !op_fused_ && arg_fused_g_ && arg_fused_f_ && owned_x1_ == 0 && opts_.allow_arg_fusion && opts_.residual_sums_in_prox > 0 &&
                 (opts_.residual_sums_in_prox >= 2 || this->problem_->ncols() + this->problem_->nrows() >= ((size_t)1 << 23));
  for (auto& p : prox_g_) res_in_prox_ = res_in_prox_ && p->supports_op_source();
  for (auto& p : prox_fstar_) res_in_prox_ = res_in_prox_ && p->supports_op_source();
  // every prox launch of a side writes its partial sums into its own run of slots of that side's half of the workspace: more proxes than
  // slots, or a prox without elements (it launches nothing, its slots would stay unwritten), keep the separate reductions
  {
    bool slots_ok = prox_g_.size() <= kOpSumSlots / 4 && prox_fstar_.size() <= kOpSumSlots / 4;
    for (auto& p : prox_g_) slots_ok = slots_ok && p->size() > 0;
    for (auto& p : prox_fstar_) slots_ok = slots_ok && p->size() > 0;
    op_fused_ = op_fused_ && slots_ok;
    res_in_prox_ = res_in_prox_ && slots_ok;
  }
  if ((op_fused_ || res_in_prox_) && !op_workspace_) {
    CheckHip(prost_hip_malloc(&op_workspace_, 2 * (size_t)kOpSumSlots * 4 * sizeof(double)), "malloc");
    CheckHip(prost_hip_memset(op_workspace_, 0, 2 * (size_t)kOpSumSlots * 4 * sizeof(double), CurrentStream()), "memset");
  }
  single_kernel_ = fused_ && opts_.allow_single_kernel && prost_hip_fused_iteration_supported(&desc_, dtype_id<T>()) == 1;
  single3d_ = fused_ && !single_kernel_ && opts_.allow_single_kernel && prost_hip_fused_iteration3d_supported(&desc_, dtype_id<T>()) == 1;
  single3d_pw_ = single3d_ && prost_hip_fused_iteration3d_pw_supported(&desc_, dtype_id<T>()) == 1;
  single_mc_ = fused_ && !single_kernel_ && !single3d_ && opts_.allow_single_kernel && prost_hip_fused_iteration_mc_supported(&desc_, dtype_id<T>()) == 1;

  x_.resize(n); x_prev_.resize(n); y_.resize(m); y_prev_.resize(m);
  if (!fused_) { kty_prev_.resize(n); kty_.resize(n); kx_.resize(m); kx_prev_.resize(m); temp_.resize(l); }
  TryMaskedPairShape();
  pair_kernel_ = single_kernel_ && opts_.allow_pair_kernel && prost_hip_fused_iteration2_profitable(&desc_pair_, dtype_id<T>()) == 1;
  // (volumes of fewer than 4 planes leave 13 of the 16 wavefronts of a workgroup idle: 2048^2 x 2 runs 0.125 ms per iteration in pairs,
  // 0.086 ms in single launches; from 4 planes on the pairs win, 0.127 against 0.206 ms)
  pair3d_ = fused_ && desc_.is3d && desc_.L >= 4 && opts_.allow_single_kernel && opts_.allow_pair_kernel && prost_hip_fused_iteration3d_x2_supported(&desc_, dtype_id<T>()) == 1;
  // 2-4 channels: the channels on the wavefronts of a workgroup, two iterations per launch
  // (also at heights the single-iteration kernels do not take: the other iterations then run the two passes)
  // (L = 2 at heights the gray-value pair kernel takes: PerformIterations launches that one, so only one of the two flags is set --
  // the kernel names and chunk lengths KernelTimes reports then belong to the kernel that ran)
  pair_mc_ = fused_ && !pair_kernel_ && !desc_.is3d && desc_.L >= 2 && desc_.L <= 4 && opts_.allow_single_kernel && opts_.allow_pair_kernel &&
             prost_hip_fused_iteration_mc_x2_profitable(&desc_pair_, dtype_id<T>()) == 1;
  // third buffers: where every residual iteration (single-kernel paths) or every other pair (pair_kernel_: stored intermediate
  // iterate) uses them they are allocated here; the 3-D / multi-channel pair paths without them need a third buffer only to
  // rebuild the previous iterate (RebuildPrevious: read-out, callbacks) and allocate it there -- 4 n + 4 m values less resident
  // at the 2048 x 2048 x 64 size until somebody reads the solution
  // tolerance-class arithmetic (Options::arithmetic): where the K-iterations-per-launch kernel takes the description, groups of up to
  // kGroupMax iterations replace the pairs; elsewhere the solve stays exact (exact results satisfy every tolerance)
  group_max_ = 0; stale_group_ = false; stale_count_ = 2;
  desc_.arith = desc_pair_.arith = PROST_HIP_ARITH_EXACT;
  if (pair_kernel_ && owned_x1_ == 0 && opts_.group_max != 1) {
    prost_hip_fused_desc probe = desc_pair_;
    probe.arith = opts_.arithmetic == PROST_HIP_ARITH_FMAD ? PROST_HIP_ARITH_FMAD : PROST_HIP_ARITH_EXACT;
    const int kmax = prost_hip_fused_iterationk_max(&probe, dtype_id<T>());
    // tolerance class: groups of up to kGroupMax.  Exact class: the K-iteration kernel with the exact forms equals the pair kernel and the
    // oracle bit for bit, but its launches are bound by instruction issue (~48 us per iteration whatever K: 4096^2, K = 2 / 3 / 4: 0.106 /
    // 0.145 / 0.204 ms per launch), so groups buy nothing there: pairs stay the default, Options::group_max > 1 asks for groups (tests)
    const int want = opts_.group_max > 1 ? opts_.group_max : (probe.arith == PROST_HIP_ARITH_FMAD ? kGroupMax : 1);
    if (kmax >= 2 && want >= 2) { group_max_ = std::min(std::min(kmax, kGroupMax), want); desc_pair_.arith = probe.arith; }
  }
  // gradient3d volumes / 2-4 channels: the pair kernels have tolerance-class instances of their own (two iterations per launch as before)
  if (opts_.arithmetic == PROST_HIP_ARITH_FMAD && pair3d_) {
    prost_hip_fused_desc probe = desc_;
    probe.arith = PROST_HIP_ARITH_FMAD;
    if (prost_hip_fused_iteration3d_x2_arith(&probe, dtype_id<T>()) == PROST_HIP_ARITH_FMAD) desc_.arith = PROST_HIP_ARITH_FMAD;
  }
  if (opts_.arithmetic == PROST_HIP_ARITH_FMAD && pair_mc_) {
    prost_hip_fused_desc probe = desc_pair_;
    probe.arith = PROST_HIP_ARITH_FMAD;
    if (prost_hip_fused_iteration_mc_x2_arith(&probe, dtype_id<T>()) == PROST_HIP_ARITH_FMAD) desc_pair_.arith = PROST_HIP_ARITH_FMAD;
  }
  if (pair_kernel_) x_spare_.resize(n);
  if (single_kernel_ || single3d_ || single_mc_) y_spare_.resize(m);

  CheckHip(prost_hip_malloc((void**)&res_dev_, 4 * sizeof(double)), "malloc");
  CheckHip(prost_hip_memset(res_dev_, 0, 4 * sizeof(double), CurrentStream()), "memset");
  CheckHip(prost_hip_host_alloc((void**)&res_host_, 4 * sizeof(double)), "host_alloc");
  CheckHip(prost_hip_malloc(&workspace_, prost_hip_reduce_workspace_bytes()), "malloc");
  side_inflight_ = resolve_on_side_ = false;
  // kernel timing: a first batch of events now, so that a short timed run does not create them (a few microseconds each) between its launches
  while (ev_.size() < 64) { void* e; CheckHip(prost_hip_event_create_timing(&e), "event_create"); ev_.push_back(e); }
  const bool deferred_rule = opts_.stepsize_variant != kPDHGStepsResidualGoldstein && opts_.stepsize_variant != kPDHGStepsResidualBoyd;
  if (this->comm_ && deferred_rule && owned_x1_ == 0) {     // column-sharded slabs also exchange halos on this communicator: keep one stream there
    CheckHip(prost_hip_stream_create(&side_stream_), "stream_create");
    CheckHip(prost_hip_event_create(&ev_res_ready_), "event_create");
    CheckHip(prost_hip_event_create(&ev_res_done_), "event_create");
  }

  // goldstein / boyd on the one-kernel 2-D path: rule + stopping test on the device (header).  Needs every step-size dependent prox
  // term to be a scalar (a, c, e of prox_g; prox_f* has scalars only on this path) and e = 0 on both sides (the kernels' dispatch
  // must not depend on the step size); column-sharded slabs exchange halos between iterations and keep the host loop.
  // (gray values, and 2-4 channels with the channels in one lane / on the wavefronts of a workgroup: every kernel of those paths
  // reads the record -- prost_hip_fused_iteration_rec, _iteration_mc_rec, _iteration2_rec, _iteration_mc_x2_rec)
  // (round 5: gradient3d as well -- prost_hip_fused_iteration3d_rec, _3d_pw_rec, _3d_x2_rec: the reference's default options on a
  // volume cost a host round trip per iteration before)
  const bool rec_kernels = single_kernel_ || single_mc_ || single3d_;
  // (column-sharded slabs exchange halos between iterations: on the RCCL transport the exchange is device-side work on the solver's
  // stream, enqueued by the exchange hook inside the batch; on the host-callback transport it needs the host and the host loop stays)
  const bool slab_ok = owned_x1_ == 0 || (this->comm_ && prost_hip_comm_is_host(this->comm_) == 0);
  dev_rules_ = rec_kernels && opts_.allow_device_rules && slab_ok && !deferred_rule && !desc_.g_coeff_ptr[0] && !desc_.g_coeff_ptr[2] &&
               !desc_.g_coeff_ptr[4] && desc_.g_coeff_val[4] == 0.0 && desc_.f_coeff_val[4] == 0.0;
  // the same on the GENERIC path (any operator): the proxes form their arguments on the fly with kernels that read tau, sigma, theta
  // from the record (elem operations of any function and coefficients, their Moreau wraps, the identity), the residual reductions
  // as well, and a one-thread kernel applies the rule behind them
  // (round 5: a prox that needs its argument materialised -- a plugin's ProxElemOperation<T, OP> -- takes part when its kernel can read the
  // step size from the device, Prox::takes_step_view; the argument pass in front of it reads the record like the fused ones)
  dev_rules_generic_ = !fused_ && opts_.allow_device_rules && owned_x1_ == 0 && !deferred_rule;
  for (auto& p : prox_g_) dev_rules_generic_ = dev_rules_generic_ && (p->supports_arg_source() ? p->takes_step_record() : p->takes_step_view());
  for (auto& p : prox_fstar_) dev_rules_generic_ = dev_rules_generic_ && (p->supports_arg_source() ? p->takes_step_record() : p->takes_step_view());
  in_device_batch_ = false; dev_batches_ = 0;
  if (dev_rules_ || dev_rules_generic_) {
    CheckHip(prost_hip_malloc(&rule_rec_, prost_hip_pdhg_rule_record_bytes()), "malloc");
    const void *vt = nullptr, *vs = nullptr;
    CheckHip(prost_hip_pdhg_record_view(rule_rec_, dtype_id<T>(), &vt, &vs, nullptr, &view_stop_), "pdhg_record_view");
    view_tau_ = static_cast<const T*>(vt); view_sigma_ = static_cast<const T*>(vs);
    CheckHip(prost_hip_host_alloc((void**)&rule_mirror_, sizeof(prost_hip_pdhg_rule_state)), "host_alloc");
    // (the rule kernels write their scalars to a DEVICE copy, fetched once per batch: ~20 stores over PCIe per residual iteration
    // cost the one-thread epilogue ~4 us -- a fifth of an iteration at 1024^2)
    CheckHip(prost_hip_malloc((void**)&rule_mirror_dev_, sizeof(prost_hip_pdhg_rule_state)), "malloc");
  }

  this->primal_var_norm_ = this->dual_var_norm_ = this->primal_residual_ = this->dual_residual_ = 0;

  if (opts_.scale_steps_operator) {                        // backend_pdhg.cu:274-286
    T norm = this->problem_->normest();
    if (std::abs(norm - 1) > 0.1) {
      tau_ /= norm;
      sigma_ /= norm;
      if (this->solver_opts_.verbose)
        std::cout << "|K|=" << norm << " => Rescaled tau=" << tau_ << ", sigma=" << sigma_ << "." << std::endl;
    }
  }
  if (this->solver_opts_.x0.size() > 0) {
    if (this->solver_opts_.x0.size() != n) throw Exception("Initial primal solution has wrong size.");
    x_ = this->solver_opts_.x0; x_prev_ = this->solver_opts_.x0;
  }
  if (this->solver_opts_.y0.size() > 0) {
    if (this->solver_opts_.y0.size() != m) throw Exception("Initial dual solution has wrong size.");
    y_ = this->solver_opts_.y0; y_prev_ = this->solver_opts_.y0;
  }
}

template <typename T>
void BackendPDHG<T>::Release() {
  if (res_dev_) { prost_hip_free(res_dev_); res_dev_ = nullptr; }
  if (res_host_) { prost_hip_host_free(res_host_); res_host_ = nullptr; }
  if (workspace_) { prost_hip_free(workspace_); workspace_ = nullptr; }
  if (op_workspace_) { prost_hip_free(op_workspace_); op_workspace_ = nullptr; }
  if (rule_rec_) { prost_hip_free(rule_rec_); rule_rec_ = nullptr; }
  if (rule_mirror_) { prost_hip_host_free(rule_mirror_); rule_mirror_ = nullptr; }
  if (rule_mirror_dev_) { prost_hip_free(rule_mirror_dev_); rule_mirror_dev_ = nullptr; }
  batch_marks_.clear();
  if (side_stream_) { prost_hip_stream_synchronize(side_stream_); prost_hip_stream_destroy(side_stream_); side_stream_ = nullptr; }
  if (ev_res_ready_) { prost_hip_event_destroy(ev_res_ready_); ev_res_ready_ = nullptr; }
  if (ev_res_done_) { prost_hip_event_destroy(ev_res_done_); ev_res_done_ = nullptr; }
  if (ev_res_local_) { prost_hip_event_destroy(ev_res_local_); ev_res_local_ = nullptr; }
  spec_valid_ = false;
  for (void* e : ev_) prost_hip_event_destroy(e);
  ev_.clear(); samples_.clear(); ev_used_ = 0; last_end_ = kNoEvent;
  y_spare_.clear(); x_spare_.clear(); sol_z_.clear(); sol_w_.clear(); b_masked_.clear();
  for (auto& v : merged_g_) v.clear();
  x_.clear(); y_.clear(); x_prev_.clear(); y_prev_.clear(); temp_.clear(); kx_.clear(); kty_.clear(); kx_prev_.clear(); kty_prev_.clear();
}

template <typename T>
void BackendPDHG<T>::PerformIteration() {
  const bool residual_iteration = is_residual_iteration(iteration_);
  if (fused_) IterationFused(residual_iteration); else IterationGeneric(residual_iteration);
}

/// Two iterations (k, k+1) in one launch whenever k >= 2 is not a residual iteration (iterations 0
/// and 1 run with zeroed K^T y / K x vectors, backend_pdhg.cu:213-216; a residual iteration as the
/// FIRST of a pair would need y^(k-1)).  The iterate in between, x^(k+1) / y^(k+1), stays in
/// registers: if k+1 is a residual iteration its sums are formed in the kernel, and if somebody
/// later asks for the previous iterate (current_solution with z / w, i.e. callbacks, convergence,
/// the end of the run) it is rebuilt by ONE single launch from the pair's inputs (RebuildPrevious).
/// Only when iteration k+2 is a residual iteration -- it streams y^(k+1) as y_prev -- does the pair
/// store the intermediate iterate itself.
template <typename T>
int BackendPDHG<T>::PerformIterations(int budget) {
  // slabs with an exchange hook: a device-resident batch calls the hook itself (PerformIterationsDevice); every other path exchanges here
  // when it is due and never runs across the next exchange
  const bool hooked = exchange_hook_ && exchange_period_ > 0;
  const bool batch = (dev_rules_ || dev_rules_generic_) && budget >= 3 && iteration_ >= 2 &&
                     !(stop_on_convergence_ && this->primal_residual_ < this->eps_primal() && this->dual_residual_ < this->eps_dual());
  if (hooked && !batch) {
    if (since_exchange_ >= exchange_period_) { exchange_hook_(); since_exchange_ = 0; }
    budget = (int)std::min<size_t>((size_t)budget, exchange_period_ - since_exchange_);
  }
  const int ran = PerformIterationsInner(budget);
  if (hooked && !batch) since_exchange_ += (size_t)ran;
  return ran;
}

template <typename T>
int BackendPDHG<T>::PerformIterationsInner(int budget) {
  if (failed_) throw Exception("BackendPDHG: an earlier batch of device-resident iterations failed half-way; the state on the device is undefined. Create a new solver.");
  const size_t k = iteration_;
  // z, w of the last read-out (n + m values; the reference keeps no such copies): large ones are released when the iteration
  // goes on, small ones stay for the next callback
  if (!sol_z_.empty() && (sol_z_.size() + sol_w_.size()) * sizeof(T) > ((size_t)1 << 30)) { sol_z_.clear(); sol_w_.clear(); }
  if (spec_valid_) {
    spec_valid_ = false;
    if (budget >= spec_count_ && k == spec_iteration_) {
      // the launch (k .. k+g-1) already ran, into the spare buffers: exchange them in, exactly the state IterationPair / IterationGroup leaves
      x_prev_.swap(x_spare_); x_.swap(x_prev_);          // x_ = x^(k+g), x_prev_ = x^k (the launch's input), spare = what x_prev_ held
      y_prev_.swap(y_spare_); y_.swap(y_prev_);
      prev_stale_ = true;
      stale_tau_ = spec_tau_[0]; stale_sigma_ = spec_sigma_[0]; stale_theta_ = spec_theta_[0];
      stale_count_ = spec_count_; stale_group_ = group_max_ >= 2;
      for (int i = 1; i < spec_count_; i++) { stale_tau_more_[i] = spec_tau_[i]; stale_sigma_more_[i] = spec_sigma_[i]; stale_theta_more_[i] = spec_theta_[i]; }
      tau_ = spec_tau_[spec_count_]; sigma_ = spec_sigma_[spec_count_]; theta_ = spec_theta_[spec_count_];
      iteration_ += (size_t)spec_count_;
      pair_launches_++;
      spec_adopted_++;
      return spec_count_;
    }
  }
  // residual-driven rules on the device: a batch of iterations, ONE host wait at its end (two- and one-iteration budgets -- a user
  // stop callback polls after every launch -- keep the host loop: the batch's set-up kernel would cost more than the wait)
  // (a caller that goes on after the stopping test already holds gets the host loop's answer -- one iteration -- not a batch that
  // stops at its first residual iteration)
  if ((dev_rules_ || dev_rules_generic_) && budget >= 3 && k >= 2 &&
      !(stop_on_convergence_ && this->primal_residual_ < this->eps_primal() && this->dual_residual_ < this->eps_dual()))
    return PerformIterationsDevice(budget);
  {
    bool res = false;
    const int g = GroupSize(k, budget, res);
    if (g >= 2) {
      IterationGroup(g, res);
      pair_launches_++;
      return g;
    }
  }
  if (pair_kernel_ && budget >= 2 && k >= 2 && !is_residual_iteration(k)) {
    IterationPair(is_residual_iteration(k + 2), is_residual_iteration(k + 1));
    pair_launches_++;
    return 2;
  }
  // gradient3d: the double-iteration kernel forms the residual sums of its second iteration but stores no intermediate
  // iterate, so it runs unless k + 2 is a residual iteration (which streams y^(k+1)): at an even residual_iter the
  // residual iterations are always the second of a pair
  if (pair3d_ && budget >= 2 && k >= 2 && !is_residual_iteration(k) && !is_residual_iteration(k + 2)) {
    IterationPair3D(is_residual_iteration(k + 1));
    pair_launches_++;
    return 2;
  }
  // 2-4 channels: the same rule (residual sums of the second iteration in the kernel, no stored intermediate iterate)
  if (pair_mc_ && budget >= 2 && k >= 2 && !is_residual_iteration(k) && !is_residual_iteration(k + 2)) {
    IterationPairMc(is_residual_iteration(k + 1));
    pair_launches_++;
    return 2;
  }
  PerformIteration();
  return 1;
}

template <typename T>
void BackendPDHG<T>::RestoreRoles(const BatchMark& m) {
  auto place = [](device_vector<T>& dst, T* want, device_vector<T>& o1, device_vector<T>& o2) {
    if (dst.data() == want) return;
    if (o1.data() == want) dst.swap(o1); else if (o2.data() == want) dst.swap(o2);
    else throw Exception("BackendPDHG: lost track of an iterate buffer.");
  };
  place(x_, m.x, x_prev_, x_spare_); place(x_prev_, m.xp, x_spare_, x_spare_);
  place(y_, m.y, y_prev_, y_spare_); place(y_prev_, m.yp, y_spare_, y_spare_);
  if (!fused_) { place(kx_, m.kx, kx_prev_, kx_prev_); place(kty_, m.kty, kty_prev_, kty_prev_); }
  prev_stale_ = m.prev_stale;
  stale_count_ = m.stale_count; stale_group_ = m.stale_group;
  iteration_ = m.iteration_after;
  pair_launches_ = m.pair_launches;
  // the launches behind the stopping iteration did no work: their (near-zero) samples and launch counts are withdrawn
  if (samples_.size() > m.samples) samples_.resize(m.samples);
  if (ev_used_ > m.ev_used) ev_used_ = m.ev_used;
  for (int kk = 0; kk < kKernelKinds; kk++) launches_[kk] = m.launches[kk];
  last_end_ = kNoEvent;
}

/// up to kDeviceBatch iterations with the step-size rule and the stopping test on the device; returns the iterations that RAN (fewer
/// than asked for when the stopping test fired: the state is then that of the stopping iteration, as if the host loop had stopped there)
template <typename T>
int BackendPDHG<T>::PerformIterationsDevice(int budget) {
  void* s = CurrentStream();
  // A full batch is a whole number of residual periods (and even): the launch pattern -- pairs whose SECOND iteration is the residual
  // one -- then continues across the batch boundary instead of starting over with a single launch, a stored-mid pair and a single
  // residual launch per batch (measured at 4096^2, boyd, residual_iter 10, batches of 128: -3 %).
  const int period = opts_.residual_iter % 2 == 0 ? opts_.residual_iter : 2 * opts_.residual_iter;
  const int full = period <= kDeviceBatch ? (kDeviceBatch / period) * period : kDeviceBatch;
  const int n = std::min(budget, full);
  const size_t k0 = iteration_;
  prost_hip_pdhg_rule_opts o;
  o.variant = opts_.stepsize_variant == kPDHGStepsResidualGoldstein ? PROST_PDHG_RULE_GOLDSTEIN : PROST_PDHG_RULE_BOYD;
  o.arg_nu = (double)opts_.arg_nu; o.arg_delta = (double)opts_.arg_delta; o.arb_delta = (double)opts_.arb_delta; o.arb_tau = (double)opts_.arb_tau;
  o.tol_abs_primal = (double)this->solver_opts_.tol_abs_primal; o.tol_abs_dual = (double)this->solver_opts_.tol_abs_dual;
  o.tol_rel_primal = (double)this->solver_opts_.tol_rel_primal; o.tol_rel_dual = (double)this->solver_opts_.tol_rel_dual;
  o.sqrt_rows = std::sqrt((double)(this->global_nrows_ ? this->global_nrows_ : this->problem_->nrows()));      // backend.hpp:71-74
  o.sqrt_cols = std::sqrt((double)(this->global_ncols_ ? this->global_ncols_ : this->problem_->ncols()));
  ResolveResiduals();                      // (sums of the host loop still in flight: the rule's input state must be final)
  spec_valid_ = false;
  prost_hip_fused_desc no_desc{};            // generic path: the record's prox terms are not read by any kernel
  no_desc.T_val = no_desc.S_val = 1.0; no_desc.g_coeff_val[0] = no_desc.f_coeff_val[0] = 1.0;
  CheckHip(Api<T>::pdhg_rule_begin(rule_rec_, &o, fused_ ? &desc_ : &no_desc, (double)tau_, (double)sigma_, (double)theta_, (double)arg_alpha_, arb_l_, arb_u_,
                                   stop_on_convergence_ ? 1 : 0, rule_mirror_dev_, s), "pdhg_rule_begin");
  batch_marks_.clear();
  batch_last_launch_evaluated_ = false;
  in_device_batch_ = true;
  if (!fused_) CheckHip(prost_hip_use_step_record(rule_rec_), "use_step_record");
  try {
    for (int done = 0; done < n;) {
      const size_t k = iteration_;
      batch_last_launch_evaluated_ = false;
      // slabs: the halo exchange when it is due (device-side, on this stream), and no launch across the next one
      if (exchange_hook_ && exchange_period_ > 0 && since_exchange_ >= exchange_period_) { exchange_hook_(); since_exchange_ = 0; }
      const int room = exchange_hook_ && exchange_period_ > 0 ? (int)std::min<size_t>((size_t)(n - done), exchange_period_ - since_exchange_) : n - done;
      int ran = 1;
      if (!fused_) {
        IterationGeneric(is_residual_iteration(k));
      } else if (bool gres = false; int g = GroupSize(k, room, gres)) {
        pair_launches_++;            // (counted first: the mark a residual launch leaves holds the count INCLUDING itself)
        IterationGroup(g, gres);
        ran = g;
      } else if (pair_kernel_ && room >= 2 && !is_residual_iteration(k)) {
        pair_launches_++;            // (counted first: the mark a residual launch leaves holds the count INCLUDING itself)
        IterationPair(is_residual_iteration(k + 2), is_residual_iteration(k + 1));
        ran = 2;
      } else if (pair_mc_ && room >= 2 && !is_residual_iteration(k) && !is_residual_iteration(k + 2)) {     // (no stored intermediate iterate)
        pair_launches_++;
        IterationPairMc(is_residual_iteration(k + 1));
        ran = 2;
      } else if (pair3d_ && room >= 2 && k >= 2 && !is_residual_iteration(k) && !is_residual_iteration(k + 2)) {
        pair_launches_++;
        IterationPair3D(is_residual_iteration(k + 1));
        ran = 2;
      } else {
        IterationFused(is_residual_iteration(k));
      }
      done += ran;
      since_exchange_ += (size_t)ran;
    }
  } catch (...) {
    // Launches of this batch are already enqueued: iteration_, the buffer roles and pair_launches_ have advanced with them while tau_ /
    // sigma_ / theta_ and the residual fields still hold the values from the start of the batch (the true ones exist only in the device
    // record, behind kernels that may have failed).  Nothing consistent can be handed out from here: the backend is marked failed
    // and every later iteration / read-out throws instead of pairing iterates with the wrong step sizes.
    in_device_batch_ = false; prost_hip_use_step_record(nullptr);
    batch_marks_.clear(); spec_valid_ = false; failed_ = true;
    throw;
  }
  in_device_batch_ = false;
  if (!fused_) CheckHip(prost_hip_use_step_record(nullptr), "use_step_record");
  dev_batches_++;
  CheckHip(prost_hip_memcpy_d2h(rule_mirror_, rule_mirror_dev_, sizeof(prost_hip_pdhg_rule_state), s), "memcpy_d2h");
  CheckHip(prost_hip_stream_synchronize(s), "stream_synchronize");          // the batch's ONE host wait
  CheckHip(prost_hip_check_last_error(), "PDHG iteration");
  const prost_hip_pdhg_rule_state& m = *rule_mirror_;
  if (m.evaluations > 0) {
    this->primal_residual_ = (T)m.primal_res; this->primal_var_norm_ = (T)m.primal_var;
    this->dual_residual_ = (T)m.dual_res; this->dual_var_norm_ = (T)m.dual_var;
    for (int i = 0; i < 4; i++) res_host_[i] = m.sums[i];
    arg_alpha_ = (T)m.arg_alpha; arb_l_ = (int)m.arb_l; arb_u_ = (int)m.arb_u;
  }
  tau_ = (T)m.tau; sigma_ = (T)m.sigma; theta_ = (T)m.theta;
  bool last_evaluated = batch_last_launch_evaluated_;
  if (m.stopped) {
    // the launches after the stopping iteration returned at once: back to the buffer roles that iteration left
    const BatchMark* mark = nullptr;
    for (const BatchMark& b : batch_marks_) if (b.iteration_after == (size_t)m.stop_iteration + 1) mark = &b;
    if (!mark) throw Exception("BackendPDHG: the device stopped at an iteration the batch did not contain.");
    RestoreRoles(*mark);
    last_evaluated = true;
  }
  // step sizes of the last launch that ran (RebuildPrevious re-runs its first iteration): the values before its rule evaluation
  if (last_evaluated) { stale_tau_ = (T)m.prev_tau; stale_sigma_ = (T)m.prev_sigma; stale_theta_ = (T)m.prev_theta; }
  else { stale_tau_ = tau_; stale_sigma_ = sigma_; stale_theta_ = theta_; }
  // (every iteration of a launch inside a batch ran with the record's values of that moment)
  for (int i = 0; i < kGroupMax; i++) { stale_tau_more_[i] = stale_tau_; stale_sigma_more_[i] = stale_sigma_; stale_theta_more_[i] = stale_theta_; }
  batch_marks_.clear();
  return (int)(iteration_ - k0);
}

template <typename T>
size_t BackendPDHG<T>::NewEvent() {
  if (ev_used_ == ev_.size()) { void* e; CheckHip(prost_hip_event_create_timing(&e), "event_create"); ev_.push_back(e); }
  return ev_used_++;
}

/// Timing of a launch (prost_hip_next_launch_events -> hipExtLaunchKernel): a sampled launch takes a START event -- a marker packet in
/// front of the kernel -- and a STOP event bound to the kernel's own command; the elapsed time between them is the kernel's duration
/// as rocprofv3 reports it (profiles/: within 1 %).  The marker costs the chain ~3.4 us per sampled launch with events created
/// without the system fence (4.6 us with default events; tools/stamp_probe.hip).  Tried in round 4: stop events only, a sample being
/// the distance between the ENDS of consecutive launches -- free, but that distance is the launch PERIOD: it contains the ~4 us the
/// device idles between two dependent launches of this kernel (barrier, cache write-back / invalidate, dispatch of 3 876 workgroups),
/// which is not part of the kernel's duration: 104.1 us against the 99.9 us rocprofv3 reports for the same run.  (hipEventRecord
/// brackets, round 2, cost two marker packets per launch and read the gap along with the kernel.)
template <typename T>
bool BackendPDHG<T>::BeginSample(int kind) {
  // (at most kMaxSamples launches are timed between two KernelTimes calls, later ones run untimed)
  if (!this->time_kernels_ || samples_.size() >= kMaxSamples) return false;
  // one launch in `sample_every_`
  if (this->sample_every_ > 1 ? (launches_[kind]++ % (size_t)this->sample_every_) != 1 : (launches_[kind]++, false)) return false;
  const size_t start = NewEvent(), end = NewEvent();
  CheckHip(prost_hip_next_launch_events(ev_[start], ev_[end]), "next_launch_events");
  samples_.push_back({kind, start, end});
  return true;
}

template <typename T>
void BackendPDHG<T>::EndSample(bool sampled) {
  if (sampled) CheckHip(prost_hip_next_launch_events(nullptr, nullptr), "next_launch_events");     // (a launch that did not take them)
}

template <typename T>
void BackendPDHG<T>::AbortSample(bool sampled) {
  if (!sampled) return;
  prost_hip_next_launch_events(nullptr, nullptr);          // (no CheckHip: an exception is already on its way)
  if (!samples_.empty()) { samples_.pop_back(); if (ev_used_ >= 2) ev_used_ -= 2; }
}

template <typename T>
void BackendPDHG<T>::IterationPair(bool store_mid, bool residuals) {
  void* s = CurrentStream();
  double tau[2], sigma[2], theta[2];
  tau[0] = (double)tau_; sigma[0] = (double)sigma_; theta[0] = (double)theta_;
  stale_tau_ = tau_; stale_sigma_ = sigma_; stale_theta_ = theta_;
  if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();          // step sizes of iteration k+1 (:483-488)
  iteration_++;
  tau[1] = (double)tau_; sigma[1] = (double)sigma_; theta[1] = (double)theta_;
  TimedLaunch(store_mid ? (residuals ? kKernelPairMidRes : kKernelPairMid) : (residuals ? kKernelPairRes : kKernelPair), [&] {
    if (in_device_batch_)            // step sizes from the device record (both iterations: no rule evaluation falls between them)
      CheckHip(Api<T>::fused_iteration2_rec(&desc_pair_, store_mid ? x_spare_.data() : x_prev_.data(), store_mid ? y_spare_.data() : y_prev_.data(), x_.data(),
                                            y_.data(), store_mid ? x_prev_.data() : nullptr, store_mid ? y_prev_.data() : nullptr, rule_rec_, 0,
                                            residuals ? res_target() : nullptr, residuals ? workspace_ : nullptr, this->comm_ ? 0 : 1,
                                            (unsigned long long)iteration_, rule_mirror_dev_, s), "fused_iteration2_rec");
    else if (!store_mid)
      CheckHip(Api<T>::fused_iteration2(&desc_pair_, x_prev_.data(), y_prev_.data(), x_.data(), y_.data(), nullptr, nullptr, tau, sigma, theta, 0,
                                        residuals ? res_target() : nullptr, residuals ? workspace_ : nullptr, s), "fused_iteration2");
    else
      CheckHip(Api<T>::fused_iteration2(&desc_pair_, x_spare_.data(), y_spare_.data(), x_.data(), y_.data(), x_prev_.data(), y_prev_.data(), tau, sigma,
                                        theta, 0, residuals ? res_target() : nullptr, residuals ? workspace_ : nullptr, s), "fused_iteration2");
  });
  stale_count_ = 2; stale_group_ = false;
  if (!store_mid) {
    x_.swap(x_prev_);        // x_ = x^(k+2); x_prev_ / y_prev_ = x^k / y^k, the pair's inputs
    y_.swap(y_prev_);
    prev_stale_ = true;
  } else {
    x_.swap(x_spare_);       // x_ = x^(k+2), x_prev_ = x^(k+1): the state two single launches leave
    y_.swap(y_spare_);
    prev_stale_ = false;
  }
  if (residuals) FinishResiduals();                                    // iteration_ == k+1 here, as in the single path
  if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
  iteration_++;
}

/// The launch PerformIterations makes at iteration k: a group never starts with a residual iteration (its sums need the iterate in
/// front of it), never contains one except as its LAST iteration (the kernel forms the sums there), and never leaves the next residual
/// iteration alone behind it (a single residual launch streams y^(k-1), which a group keeps in registers): with d = iterations up to and
/// including the next residual one, g = min(budget, group_max_, d), shortened while that would leave d - g == 1 or 2.
/// workgroups of one residual launch of the generic path (kOpLaunchSlots; PROST_OP_LAUNCH_SLOTS overrides it for measurements)
static unsigned OpLaunchSlots() {
  static const unsigned v = []() { const char* e = getenv("PROST_OP_LAUNCH_SLOTS"); return e && atoi(e) > 0 ? (unsigned)atoi(e) : 2048u; }();
  return v;
}

template <typename T>
int BackendPDHG<T>::GroupSize(size_t k, int budget, bool& residuals) const {
  residuals = false;
  if (group_max_ < 2 || budget < 2 || k < 2 || is_residual_iteration(k)) return 0;
  const size_t ri = (size_t)opts_.residual_iter;
  const size_t r = (k / ri + 1) * ri;                // the next residual iteration (> k)
  const size_t d = r - k + 1;
  int g = (int)std::min<size_t>(std::min<size_t>((size_t)budget, (size_t)group_max_), d);
  // never leave one iteration behind (see above), and not two either where a shorter launch now avoids it: the two-iteration launch
  // with the sums is the slowest instance per iteration (4096^2: 0.101-0.127 ms between boxes against 0.104 for three), so a period of
  // ten runs as 4 + 3 + 3, not 4 + 4 + 2
  while (g > 2 && d - (size_t)g >= 1 && d - (size_t)g <= 2) g--;
  residuals = k + (size_t)g - 1 == r;
  return g;
}

template <typename T>
void BackendPDHG<T>::IterationGroup(int g, bool residuals) {
  void* s = CurrentStream();
  double tau[kGroupMax], sigma[kGroupMax], theta[kGroupMax];
  for (int i = 0; i < g; i++) {
    tau[i] = (double)tau_; sigma[i] = (double)sigma_; theta[i] = (double)theta_;
    if (i == 0) { stale_tau_ = tau_; stale_sigma_ = sigma_; stale_theta_ = theta_; }
    else { stale_tau_more_[i] = tau_; stale_sigma_more_[i] = sigma_; stale_theta_more_[i] = theta_; }
    if (i + 1 < g) {
      if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();          // step sizes of the next iteration (:483-488)
      iteration_++;
    }
  }
  // iteration_ == k + g - 1 here, the index of the launch's last iteration, as FinishResiduals expects it
  const int kind = (residuals ? kKernelGroup2Res : kKernelGroup2) + (g - 2);
  TimedLaunch(kind, [&] {
    if (in_device_batch_)            // step sizes from the device record (every iteration of the launch: no rule evaluation falls between them)
      CheckHip(Api<T>::fused_iterationk_rec(&desc_pair_, g, x_prev_.data(), y_prev_.data(), x_.data(), y_.data(), rule_rec_, 0, residuals ? res_target() : nullptr,
                                            residuals ? workspace_ : nullptr, this->comm_ ? 0 : 1, (unsigned long long)iteration_, rule_mirror_dev_, s), "fused_iterationk_rec");
    else
      CheckHip(Api<T>::fused_iterationk(&desc_pair_, g, x_prev_.data(), y_prev_.data(), x_.data(), y_.data(), tau, sigma, theta, 0,
                                        residuals ? res_target() : nullptr, residuals ? workspace_ : nullptr, s), "fused_iterationk");
  });
  x_.swap(x_prev_);        // x_ = x^(k+g); x_prev_ / y_prev_ = x^k / y^k, the launch's inputs
  y_.swap(y_prev_);
  prev_stale_ = true; stale_count_ = g; stale_group_ = true;
  if (residuals) FinishResiduals();
  if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
  iteration_++;
}

template <typename T>
void BackendPDHG<T>::IterationPair3D(bool residuals) {
  double tau[2], sigma[2], theta[2];
  tau[0] = (double)tau_; sigma[0] = (double)sigma_; theta[0] = (double)theta_;
  stale_tau_ = tau_; stale_sigma_ = sigma_; stale_theta_ = theta_;
  if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();          // step sizes of iteration k+1 (:483-488)
  iteration_++;
  tau[1] = (double)tau_; sigma[1] = (double)sigma_; theta[1] = (double)theta_;
  TimedLaunch(residuals ? kKernelPairRes : kKernelPair, [&] {
    if (in_device_batch_)
      CheckHip(Api<T>::fused_iteration3d_x2_rec(&desc_, x_prev_.data(), y_prev_.data(), x_.data(), y_.data(), rule_rec_, 0, residuals ? res_target() : nullptr,
                                                residuals ? workspace_ : nullptr, this->comm_ ? 0 : 1, (unsigned long long)iteration_, rule_mirror_dev_, CurrentStream()),
               "fused_iteration3d_x2_rec");
    else
    CheckHip(Api<T>::fused_iteration3d_x2(&desc_, x_prev_.data(), y_prev_.data(), x_.data(), y_.data(), tau, sigma, theta, 0,
                                          residuals ? res_target() : nullptr, residuals ? workspace_ : nullptr, CurrentStream()), "fused_iteration3d_x2");
  });
  x_.swap(x_prev_);        // x_ = x^(k+2); x_prev_ / y_prev_ = x^k / y^k, the pair's inputs
  y_.swap(y_prev_);
  prev_stale_ = true; stale_count_ = 2; stale_group_ = false;
  if (residuals) FinishResiduals();                                    // iteration_ == k+1 here, as in the single path
  if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
  iteration_++;
}

template <typename T>
void BackendPDHG<T>::IterationPairMc(bool residuals) {
  double tau[2], sigma[2], theta[2];
  tau[0] = (double)tau_; sigma[0] = (double)sigma_; theta[0] = (double)theta_;
  stale_tau_ = tau_; stale_sigma_ = sigma_; stale_theta_ = theta_;
  if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();          // step sizes of iteration k+1 (:483-488)
  iteration_++;
  tau[1] = (double)tau_; sigma[1] = (double)sigma_; theta[1] = (double)theta_;
  TimedLaunch(residuals ? kKernelPairRes : kKernelPair, [&] {
    if (in_device_batch_)
      CheckHip(Api<T>::fused_iteration_mc_x2_rec(&desc_pair_, x_prev_.data(), y_prev_.data(), x_.data(), y_.data(), rule_rec_, 0, residuals ? res_target() : nullptr,
                                                 residuals ? workspace_ : nullptr, this->comm_ ? 0 : 1, (unsigned long long)iteration_, rule_mirror_dev_, CurrentStream()),
               "fused_iteration_mc_x2_rec");
    else
    CheckHip(Api<T>::fused_iteration_mc_x2(&desc_pair_, x_prev_.data(), y_prev_.data(), x_.data(), y_.data(), tau, sigma, theta, 0,
                                           residuals ? res_target() : nullptr, residuals ? workspace_ : nullptr, CurrentStream()), "fused_iteration_mc_x2");
  });
  x_.swap(x_prev_);        // x_ = x^(k+2); x_prev_ / y_prev_ = x^k / y^k, the pair's inputs
  y_.swap(y_prev_);
  prev_stale_ = true; stale_count_ = 2; stale_group_ = false;
  if (residuals) FinishResiduals();                                    // iteration_ == k+1 here, as in the single path
  if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
  iteration_++;
}

/// x_prev_ / y_prev_ hold x^k, y^k (inputs of the last pair launch), x_ / y_ = x^(k+2), y^(k+2): one
/// single-iteration launch with the step sizes of iteration k rebuilds x^(k+1), y^(k+1) bit for bit.
template <typename T>
void BackendPDHG<T>::RebuildPrevious() {
  spec_valid_ = false;               // (the spare buffers are about to be written, or the state to be looked at)
  if (!prev_stale_) return;
  last_end_ = kNoEvent;
  if (x_spare_.size() != x_.size()) x_spare_.resize(x_.size());
  if (y_spare_.size() != y_.size()) y_spare_.resize(y_.size());
  if (stale_group_) {
    // a group of stale_count_ iterations: the iterate in front of its last one is stale_count_ - 1 iterations behind the inputs, in the
    // same arithmetic (a launch of K iterations equals any partition of it into shorter launches bit for bit)
    double tau[kGroupMax], sigma[kGroupMax], theta[kGroupMax];
    tau[0] = (double)stale_tau_; sigma[0] = (double)stale_sigma_; theta[0] = (double)stale_theta_;
    for (int i = 1; i < stale_count_ - 1; i++) { tau[i] = (double)stale_tau_more_[i]; sigma[i] = (double)stale_sigma_more_[i]; theta[i] = (double)stale_theta_more_[i]; }
    CheckHip(Api<T>::fused_iterationk(&desc_pair_, stale_count_ - 1, x_spare_.data(), y_spare_.data(), x_prev_.data(), y_prev_.data(), tau, sigma, theta, 0,
                                      nullptr, nullptr, CurrentStream()), "fused_iterationk");
  } else
  if (pair_mc_ && single_mc_)
    CheckHip(Api<T>::fused_iteration_mc(&desc_, x_spare_.data(), y_spare_.data(), x_prev_.data(), y_prev_.data(), nullptr, (double)stale_tau_,
                                        (double)stale_sigma_, (double)stale_theta_, 1, 1, 1, 0, nullptr, nullptr, CurrentStream()), "fused_iteration_mc");
  else if (pair3d_ && single3d_)
    CheckHip(Api<T>::fused_iteration3d(&desc_, x_spare_.data(), y_spare_.data(), x_prev_.data(), y_prev_.data(), nullptr, (double)stale_tau_,
                                       (double)stale_sigma_, (double)stale_theta_, 1, 1, 1, 0, nullptr, nullptr, CurrentStream()), "fused_iteration3d");
  else if (pair3d_ || (pair_mc_ && !single_kernel_)) {        // heights the one-kernel iterations do not take: the two passes
    CheckHip(Api<T>::fused_primal(&desc_, x_spare_.data(), x_prev_.data(), y_prev_.data(), nullptr, (double)stale_tau_, 1, 0, nullptr, workspace_,
                                  CurrentStream()), "fused_primal");
    CheckHip(Api<T>::fused_dual(&desc_, y_spare_.data(), y_prev_.data(), x_spare_.data(), x_prev_.data(), (double)stale_sigma_, (double)stale_theta_, 1,
                                nullptr, workspace_, CurrentStream()), "fused_dual");
  } else
  CheckHip(Api<T>::fused_iteration(&desc_, x_spare_.data(), y_spare_.data(), x_prev_.data(), y_prev_.data(), nullptr, (double)stale_tau_,
                                   (double)stale_sigma_, (double)stale_theta_, 1, 1, 1, 0, nullptr, nullptr, CurrentStream()), "fused_iteration");
  x_prev_.swap(x_spare_);
  y_prev_.swap(y_spare_);
  prev_stale_ = false;
}

/// two kernels: x_ / y_ ping-pong with x_prev_ / y_prev_
template <typename T>
void BackendPDHG<T>::IterationFused(bool res) {
  void* s = CurrentStream();
  // at entry: x_ = x^k, y_ = y^k, y_prev_ = y^(k-1).  The reference's kty_ is K^T y^k except at
  // k = 0 (zero vector, :213); kty_prev_ is K^T y^(k-1) except at k <= 1 (zero vector).
  if (single_kernel_) {
    if (res) RebuildPrevious();      // the residual kernel streams y^(k-1)
    // ONE kernel per iteration, x_new never round-trips through HBM (7 floats/pixel; residual
    // iterations add the y_prev stream and the four residual sums).  y_new cannot overwrite
    // y_prev_ on residual iterations (the kernel still reads it), so it goes to y_spare_.
    T* y_out = res ? y_spare_.data() : y_prev_.data();
    TimedLaunch(res ? kKernelIterRes : kKernelIter, [&] {
      if (in_device_batch_)
        CheckHip(Api<T>::fused_iteration_rec(&desc_, x_prev_.data(), y_out, x_.data(), y_.data(), res ? y_prev_.data() : nullptr, rule_rec_,
                                             iteration_ >= 1 ? 1 : 0, iteration_ >= 1 ? 1 : 0, iteration_ >= 2 ? 1 : 0, 0, res ? res_target() : nullptr,
                                             res ? workspace_ : nullptr, this->comm_ ? 0 : 1, (unsigned long long)iteration_, rule_mirror_dev_, s), "fused_iteration_rec");
      else
      CheckHip(Api<T>::fused_iteration(&desc_, x_prev_.data(), y_out, x_.data(), y_.data(), res ? y_prev_.data() : nullptr, (double)tau_,
                                       (double)sigma_, (double)theta_, iteration_ >= 1 ? 1 : 0, iteration_ >= 1 ? 1 : 0,
                                       iteration_ >= 2 ? 1 : 0, 0, res ? res_target() : nullptr, res ? workspace_ : nullptr, s), "fused_iteration");
    });
    x_.swap(x_prev_);
    if (res) { y_prev_.swap(y_spare_); }     // y_prev_ now holds y^(k+1); swapped into y_ below
    y_.swap(y_prev_);
    prev_stale_ = false;
    if (res) FinishResiduals();
    if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
    iteration_++;
    return;
  }
  if (single3d_) {
    // gradient3d: one kernel, x_new stays in registers (9 instead of 14 values per voxel; residual iterations add the
    // y_prev stream of the own plane and the four sums).  Outputs go to the previous-iterate buffers, which a
    // non-residual iteration does not read; on residual iterations y_new goes to y_spare_ (the kernel still reads y_prev_).
    if (res) RebuildPrevious();      // the residual kernel streams y^(k-1)
    T* y_out = res ? y_spare_.data() : y_prev_.data();
    TimedLaunch(res ? kKernelIterRes : kKernelIter, [&] {
      if (in_device_batch_ && single3d_pw_ && !res)
        CheckHip(Api<T>::fused_iteration3d_pw_rec(&desc_, x_prev_.data(), y_out, x_.data(), y_.data(), rule_rec_, iteration_ >= 1 ? 1 : 0, iteration_ >= 1 ? 1 : 0, 0, 0, s),
                 "fused_iteration3d_pw_rec");
      else if (in_device_batch_)
        CheckHip(Api<T>::fused_iteration3d_rec(&desc_, x_prev_.data(), y_out, x_.data(), y_.data(), res ? y_prev_.data() : nullptr, rule_rec_, iteration_ >= 1 ? 1 : 0,
                                               iteration_ >= 1 ? 1 : 0, iteration_ >= 2 ? 1 : 0, 0, res ? res_target() : nullptr, res ? workspace_ : nullptr,
                                               this->comm_ ? 0 : 1, (unsigned long long)iteration_, rule_mirror_dev_, s), "fused_iteration3d_rec");
      else if (single3d_pw_ && !res)        // planes across the wavefronts of a workgroup: x_new of the plane above comes through LDS
        CheckHip(Api<T>::fused_iteration3d_pw(&desc_, x_prev_.data(), y_out, x_.data(), y_.data(), (double)tau_, (double)sigma_, (double)theta_,
                                              iteration_ >= 1 ? 1 : 0, iteration_ >= 1 ? 1 : 0, 0, 0, s), "fused_iteration3d_pw");
      else
      CheckHip(Api<T>::fused_iteration3d(&desc_, x_prev_.data(), y_out, x_.data(), y_.data(), res ? y_prev_.data() : nullptr, (double)tau_, (double)sigma_,
                                         (double)theta_, iteration_ >= 1 ? 1 : 0, iteration_ >= 1 ? 1 : 0, iteration_ >= 2 ? 1 : 0, 0,
                                         res ? res_target() : nullptr, res ? workspace_ : nullptr, s), "fused_iteration3d");
    });
    x_.swap(x_prev_);
    if (res) y_prev_.swap(y_spare_);
    y_.swap(y_prev_);
    prev_stale_ = false;
    if (res) FinishResiduals();
    if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
    iteration_++;
    return;
  }
  if (single_mc_) {
    // gradient2d with 3 / 4 channels: one kernel, the channels on the wavefronts of a workgroup (7 instead of 11 values per
    // pixel and channel; residual iterations add the y_prev stream and the four sums, y_new then goes to y_spare_)
    if (res) RebuildPrevious();      // the residual kernel streams y^(k-1)
    T* y_out = res ? y_spare_.data() : y_prev_.data();
    TimedLaunch(res ? kKernelIterRes : kKernelIter, [&] {
      if (in_device_batch_)
        CheckHip(Api<T>::fused_iteration_mc_rec(&desc_, x_prev_.data(), y_out, x_.data(), y_.data(), res ? y_prev_.data() : nullptr, rule_rec_,
                                                iteration_ >= 1 ? 1 : 0, iteration_ >= 1 ? 1 : 0, iteration_ >= 2 ? 1 : 0, 0, res ? res_target() : nullptr,
                                                res ? workspace_ : nullptr, this->comm_ ? 0 : 1, (unsigned long long)iteration_, rule_mirror_dev_, s), "fused_iteration_mc_rec");
      else
      CheckHip(Api<T>::fused_iteration_mc(&desc_, x_prev_.data(), y_out, x_.data(), y_.data(), res ? y_prev_.data() : nullptr, (double)tau_, (double)sigma_,
                                          (double)theta_, iteration_ >= 1 ? 1 : 0, iteration_ >= 1 ? 1 : 0, iteration_ >= 2 ? 1 : 0, 0,
                                          res ? res_target() : nullptr, res ? workspace_ : nullptr, s), "fused_iteration_mc");
    });
    x_.swap(x_prev_);
    if (res) y_prev_.swap(y_spare_);
    y_.swap(y_prev_);
    prev_stale_ = false;
    if (res) FinishResiduals();
    if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
    iteration_++;
    return;
  }
  if (res) RebuildPrevious();        // the residual primal pass streams y^(k-1)
  prev_stale_ = false;
  TimedLaunch(kKernelPrimal, [&] {
    CheckHip(Api<T>::fused_primal(&desc_, x_prev_.data(), x_.data(), y_.data(), y_prev_.data(), (double)tau_, iteration_ >= 1 ? 1 : 0,
                                  iteration_ >= 2 ? 1 : 0, res ? res_target() + 2 : nullptr, workspace_, s), "fused_primal");
  });
  x_.swap(x_prev_);                        // x_ = x^(k+1), x_prev_ = x^k       (:334)
  // kx_prev_ of the reference is K x^k except at k = 0 (zero vector, :216)
  TimedLaunch(kKernelDual, [&] {
    CheckHip(Api<T>::fused_dual(&desc_, y_prev_.data(), y_.data(), x_.data(), x_prev_.data(), (double)sigma_, (double)theta_,
                                iteration_ >= 1 ? 1 : 0, res ? res_target() : nullptr, workspace_, s), "fused_dual");
  });
  y_.swap(y_prev_);                        // y_ = y^(k+1), y_prev_ = y^k       (:366)
  if (res) FinishResiduals();
  if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
  iteration_++;
}

/// The operator as a table of sparse (CSR arrays or row patterns) and gradient blocks for the operator sources of the prox kernels
/// (prost_hip_prox_elem_arg, PROST_ARG_PDHG_PRIMAL_OP / _DUAL_OP); plugin blocks, diags, Kronecker blocks, label_first gradients and a
/// dualized operator keep the separate products.
template <typename T>
bool BackendPDHG<T>::DescribeGenericOperator(bool stencils_only) {
  auto linop = this->problem_->linop();
  if (dynamic_cast<DualLinearOperator<T>*>(linop.get())) return false;
  const auto& blocks = linop->blocks();
  if (blocks.empty() || blocks.size() > (size_t)PROST_HIP_OP_MAX_BLOCKS) return false;
  prost_hip_fused_op op;
  std::memset(&op, 0, sizeof(op));
  for (const auto& b : blocks) {
    BlockDesc bd;
    if (!b->describe(bd)) return false;
    prost_hip_op_block& o = op.block[op.nblocks++];
    o.row = b->row(); o.col = b->col(); o.nrows = b->nrows(); o.ncols = b->ncols();
    if (bd.kind == BlockDesc::kSparse) {
      o.kind = PROST_OP_CSR;
      // (a stencil written out row by row: K and K^T run from row patterns)
      if (stencils_only && !(bd.ids && bd.ids_t)) return false;
      // (CSR rows without patterns are walked lane by lane inside the prox launches: beyond ~6 entries per row that is an order of magnitude
      // slower than the stand-alone product with its cooperating lanes -- example_deblurring.m's 15-tap motion blur: 367 against 4 855
      // iterations/s at 512 x 512 x 3 -- so such operators keep the separate products under every option)
      if (!(bd.ids && bd.ids_t) && bd.nnz > 6 * std::min(b->nrows(), b->ncols())) return false;
      o.val = bd.val; o.ptr = bd.ptr; o.ind = bd.ind; o.val_t = bd.val_t; o.ptr_t = bd.ptr_t; o.ind_t = bd.ind_t;
      o.ids = bd.ids; o.pptr = bd.pptr; o.rel = bd.rel; o.pval = bd.pval; o.ids_t = bd.ids_t; o.pptr_t = bd.pptr_t; o.rel_t = bd.rel_t; o.pval_t = bd.pval_t;
      o.anchor = bd.anchor; o.anchor_t = bd.anchor_t;
    } else if ((bd.kind == BlockDesc::kGradient2D || bd.kind == BlockDesc::kGradient3D) && !bd.label_first) {
      o.kind = bd.kind == BlockDesc::kGradient2D ? PROST_OP_GRAD2D : PROST_OP_GRAD3D;
      o.nx = bd.nx; o.ny = bd.ny; o.L = bd.L;
    } else {
      return false;
    }
  }
  if (prost_hip_prox_elem_arg_op_supported(&op, this->problem_->nrows(), this->problem_->ncols(), dtype_id<T>()) != 1) return false;
  gen_op_ = op;
  return true;
}

/// One generic iteration with the operator INSIDE the prox kernels: the launches of prox_g form x - tau T K^T y for their own elements
/// (and store K^T y, n values, for the next iteration's dual residual), the launches of prox_f* form y + sigma Sigma ((1 + theta) K x -
/// theta K x_prev); on residual iterations they add up the residual terms as well and ONE more launch folds the sums (+ the rule inside a
/// device batch).  K x is never written, the fill passes and the four operator launches are gone: deblurring's shape 9 -> 4 launches.
/// Same expressions, same block order, same bits as IterationGeneric (the reference's zero vectors of iterations 0 / 1 included).
template <typename T>
void BackendPDHG<T>::IterationGenericOp(bool res) {
  void* s = CurrentStream();
  const size_t n = this->problem_->ncols(), m = this->problem_->nrows();
  const device_vector<T>& Tr = this->problem_->scaling_right();
  const device_vector<T>& Sl = this->problem_->scaling_left();
  // the reduction workspace as slots of 4 doubles: primal sums (prox_f* launches) in the first half, dual sums (prox_g launches) in the second
  const unsigned half = kOpSumSlots;
  double* ws_p = static_cast<double*>(op_workspace_);
  double* ws_d = ws_p + 4 * (size_t)half;
  unsigned slot_p = 0, slot_d = 0;
  kty_.swap(kty_prev_);                    // kty_prev_ = K^T y^(k-1) as the previous iteration stored it (zeros before); kty_ is written below
  x_.swap(x_prev_);
  {
    typename Prox<T>::ArgSource src{PROST_ARG_PDHG_PRIMAL_OP, {x_prev_.data(), Tr.data(), nullptr, kty_prev_.data()}, {tau_, (T)0}};
    src.op = &gen_op_; src.op_rows = m; src.op_cols = n;
    src.w[0] = y_.data(); src.kty_out = kty_.data();
    src.use[0] = iteration_ >= 1 ? 1 : 0;                    // kty_ is the zero vector in iteration 0 (backend_pdhg.cu:213)
    if (res) { src.res_ws = ws_d; src.res_slot = &slot_d; src.res_slots_max = std::max<unsigned>(1, std::min<unsigned>(OpLaunchSlots(), half / (unsigned)std::max<size_t>(prox_g_.size(), 1))); }
    for (auto& p : prox_g_) p->EvalFromSource(x_, src, Tr, tau_);
  }
  y_.swap(y_prev_);
  {
    typename Prox<T>::ArgSource src{PROST_ARG_PDHG_DUAL_OP, {y_prev_.data(), Sl.data(), nullptr, nullptr}, {sigma_, theta_}};
    src.op = &gen_op_; src.op_rows = m; src.op_cols = n;
    src.w[0] = x_.data(); src.w[1] = x_prev_.data();
    src.use[1] = iteration_ >= 1 ? 1 : 0;                    // kx_prev_ is the zero vector in iteration 0 (:216)
    if (res) { src.res_ws = ws_p; src.res_slot = &slot_p; src.res_slots_max = std::max<unsigned>(1, std::min<unsigned>(OpLaunchSlots(), half / (unsigned)std::max<size_t>(prox_fstar_.size(), 1))); }
    for (auto& p : prox_fstar_) p->EvalFromSource(y_, src, Sl, sigma_);
  }
  if (res) {
    const bool rule_here = in_device_batch_ && !this->comm_;
    CheckHip(Api<T>::pdhg_fold_sums(res_target(), ws_p, slot_p, ws_d, slot_d, rule_here ? rule_rec_ : nullptr, rule_here ? 1 : 0, (unsigned long long)iteration_,
                                    rule_here ? rule_mirror_dev_ : nullptr, s), "pdhg_fold_sums");
    FinishResiduals();
  }
  if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
  iteration_++;
}

template <typename T>
void BackendPDHG<T>::IterationGeneric(bool res) {
  if (op_fused_) { IterationGenericOp(res); return; }
  void* s = CurrentStream();
  const size_t n = this->problem_->ncols(), m = this->problem_->nrows();
  const device_vector<T>& Tr = this->problem_->scaling_right();
  const device_vector<T>& Sl = this->problem_->scaling_left();
  // When every prox of a list can form its argument on the fly (elem operations, their conjugates, the identity),
  // the argument pass of :317-331 / :349-364 is folded into the prox kernels: same expressions, one vector less
  // written and re-read per prox.
  // (a list with both kinds -- in-tree operations next to a plugin's -- runs the argument pass once for the proxes that need it; inside a
  // device batch those take the step size from the record through a StepView)
  const bool res_here = res && res_in_prox_;
  const unsigned half = kOpSumSlots;
  double* ws_p = static_cast<double*>(op_workspace_);
  double* ws_d = ws_p ? ws_p + 4 * (size_t)half : nullptr;
  unsigned slot_p = 0, slot_d = 0;
  if (arg_fused_g_) {
    x_.swap(x_prev_);
    typename Prox<T>::ArgSource src{PROST_ARG_PDHG_PRIMAL, {x_prev_.data(), Tr.data(), kty_.data(), nullptr}, {tau_, (T)0}};
    if (res_here) {                                    // dual residual terms: x_prev, x, T, K^T y^(k-1), K^T y^k
      src.v[3] = kty_prev_.data();
      src.res_ws = ws_d; src.res_slot = &slot_d; src.res_slots_max = std::max<unsigned>(1, std::min<unsigned>(OpLaunchSlots(), half / (unsigned)std::max<size_t>(prox_g_.size(), 1)));
    }
    for (auto& p : prox_g_) p->EvalFromSource(x_, src, Tr, tau_);
  } else {
    CheckHip(Api<T>::pdhg_primal_arg(temp_.data(), x_.data(), Tr.data(), kty_.data(), (double)tau_, n, s), "primal_arg");   // :317-331
    x_.swap(x_prev_);
    const typename Prox<T>::ArgSource src{PROST_ARG_PDHG_PRIMAL, {x_prev_.data(), Tr.data(), kty_.data(), nullptr}, {tau_, (T)0}};
    for (auto& p : prox_g_) {
      if (in_device_batch_ && p->supports_arg_source()) p->EvalFromSource(x_, src, Tr, tau_);
      else if (in_device_batch_) p->EvalWithStepView(x_, temp_, Tr, typename Prox<T>::StepView{view_tau_, view_stop_});
      else p->Eval(x_, temp_, Tr, tau_);
    }
  }
  kx_.swap(kx_prev_);
  this->problem_->linop()->Eval(kx_, x_);
  if (arg_fused_f_) {
    y_.swap(y_prev_);
    typename Prox<T>::ArgSource src{PROST_ARG_PDHG_DUAL, {y_prev_.data(), Sl.data(), kx_.data(), kx_prev_.data()}, {sigma_, theta_}};
    if (res_here) {                                    // primal residual terms: y_prev, y, Sigma, K x_prev, K x
      src.res_ws = ws_p; src.res_slot = &slot_p; src.res_slots_max = std::max<unsigned>(1, std::min<unsigned>(OpLaunchSlots(), half / (unsigned)std::max<size_t>(prox_fstar_.size(), 1)));
    }
    for (auto& p : prox_fstar_) p->EvalFromSource(y_, src, Sl, sigma_);
  } else {
    CheckHip(Api<T>::pdhg_dual_arg(temp_.data(), y_.data(), Sl.data(), kx_.data(), kx_prev_.data(), (double)sigma_, (double)theta_, m, s), "dual_arg");   // :349-364
    y_.swap(y_prev_);
    const typename Prox<T>::ArgSource src{PROST_ARG_PDHG_DUAL, {y_prev_.data(), Sl.data(), kx_.data(), kx_prev_.data()}, {sigma_, theta_}};
    for (auto& p : prox_fstar_) {
      if (in_device_batch_ && p->supports_arg_source()) p->EvalFromSource(y_, src, Sl, sigma_);
      else if (in_device_batch_) p->EvalWithStepView(y_, temp_, Sl, typename Prox<T>::StepView{view_sigma_, view_stop_});
      else p->Eval(y_, temp_, Sl, sigma_);
    }
  }
  if (res) {                                                                                                   // :392-431
    // both reductions in one launch, their folds -- and inside a device batch without a communicator the rule -- in a second
    // (same sums, bit for bit, as pdhg_residual_primal + pdhg_residual_dual: five launches)
    const bool rule_here = in_device_batch_ && !this->comm_;
    if (res_here)
      CheckHip(Api<T>::pdhg_fold_sums(res_target(), ws_p, slot_p, ws_d, slot_d, rule_here ? rule_rec_ : nullptr, rule_here ? 1 : 0, (unsigned long long)iteration_,
                                      rule_here ? rule_mirror_dev_ : nullptr, s), "pdhg_fold_sums");
    else
    CheckHip(Api<T>::pdhg_residuals(res_target(), y_prev_.data(), y_.data(), Sl.data(), kx_prev_.data(), kx_.data(), (double)sigma_, (double)theta_, m, x_prev_.data(), x_.data(),
                                    Tr.data(), kty_prev_.data(), kty_.data(), (double)tau_, n, workspace_, rule_here ? rule_rec_ : nullptr, rule_here ? 1 : 0,
                                    (unsigned long long)iteration_, rule_here ? rule_mirror_dev_ : nullptr, s), "pdhg_residuals");
    FinishResiduals();
  }
  if (opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
  iteration_++;
  kty_.swap(kty_prev_);
  this->problem_->linop()->EvalAdjoint(kty_, y_);                                                              // :377-380
}

/// where the reduction kernels put the four sums: the pinned (device-visible) host buffer, or the device
/// buffer when an RCCL all-reduce has to run on them first.  Called right before a residual launch is
/// enqueued: if the previous all-reduce is still in flight on the side stream, the iteration stream is
/// made to wait for it (device-side) before the buffer is overwritten.
template <typename T>
double* BackendPDHG<T>::res_target() {
  if (in_device_batch_) return res_dev_;            // consumed on the device (all-reduce, rule kernel)
  if (!this->comm_) return res_host_;
  if (side_inflight_) {
    CheckHip(prost_hip_stream_wait_event(CurrentStream(), ev_res_done_), "stream_wait_event");
    side_inflight_ = false;
  }
  return res_dev_;
}

/// Residual iteration, device side done: all-reduce the four sums if there is a communicator and bring them
/// to the host.  The residual-driven step rules (goldstein, boyd) need the values before the next launch;
/// alg1 / alg2 do not, so there the host does NOT wait: the sums are picked up (ResolveResiduals) when
/// somebody asks for a residual -- Solver::Solve after every iteration (same behaviour as before),
/// Solver::Iterate never, solver_state / current_solution at the end.
template <typename T>
void BackendPDHG<T>::FinishResiduals() {
  void* s = CurrentStream();
  last_end_ = kNoEvent;            // kernel timing: the fold (and what follows here) sits between this launch and the next
  if (in_device_batch_) {
    // the sums stay on the device.  Without a communicator the kernel that folded them has already evaluated the rule and the stopping
    // test (fold4_rule_kernel); with one they pass the all-reduce first and a one-thread kernel follows.  Scalars are mirrored to pinned
    // host memory either way.
    if (this->comm_) {
      CheckHip(prost_hip_allreduce_sum_f64(this->comm_, res_dev_, 4, s), "allreduce");
      CheckHip(Api<T>::pdhg_rule_apply(rule_rec_, res_dev_, (unsigned long long)iteration_, rule_mirror_dev_, s), "pdhg_rule_apply");
    }
    // (generic path: IterationGeneric exchanges kty_ / kty_prev_ AFTER this call -- the mark holds the roles the iteration leaves)
    // (operator inside the prox kernels: kty_ / kty_prev_ were exchanged at the START of the iteration and stay)
    batch_marks_.push_back({iteration_ + 1, pair_launches_, x_.data(), x_prev_.data(), y_.data(), y_prev_.data(), prev_stale_,
                            kx_.data(), kx_prev_.data(), op_fused_ ? kty_.data() : kty_prev_.data(), op_fused_ ? kty_prev_.data() : kty_.data()});
    // kernel timing: launches enqueued behind a stopping iteration return at once -- their samples must not enter the averages
    // (RestoreRoles drops everything recorded after the mark it returns to)
    static_assert(kKernelKinds == 14, "BatchMark::launches holds one counter per kernel kind");
    batch_marks_.back().samples = samples_.size(); batch_marks_.back().ev_used = ev_used_;
    batch_marks_.back().stale_count = stale_count_; batch_marks_.back().stale_group = stale_group_;
    for (int kk = 0; kk < kKernelKinds; kk++) batch_marks_.back().launches[kk] = launches_[kk];
    batch_last_launch_evaluated_ = true;
    return;
  }
  // without a communicator the reduction kernels wrote the four sums straight into the pinned host buffer
  // (device-visible): no D2H copy, a stream synchronisation is all that is needed before reading them
  if (this->comm_) {
    if (side_stream_) {
      // alg1 / alg2: nothing on the iteration stream depends on the global sums, so the collective (which
      // waits for the slowest rank) and the copy run beside the next iterations instead of between them
      CheckHip(prost_hip_event_record(ev_res_ready_, s), "event_record");
      CheckHip(prost_hip_stream_wait_event(side_stream_, ev_res_ready_), "stream_wait_event");
      CheckHip(prost_hip_allreduce_sum_f64(this->comm_, res_dev_, 4, side_stream_), "allreduce");
      CheckHip(prost_hip_memcpy_d2h(res_host_, res_dev_, 4 * sizeof(double), side_stream_), "memcpy_d2h");
      CheckHip(prost_hip_event_record(ev_res_done_, side_stream_), "event_record");
      side_inflight_ = resolve_on_side_ = true;
    } else {
      CheckHip(prost_hip_allreduce_sum_f64(this->comm_, res_dev_, 4, s), "allreduce");
      CheckHip(prost_hip_memcpy_d2h(res_host_, res_dev_, 4 * sizeof(double), s), "memcpy_d2h");
    }
  }
  else if (opts_.allow_speculation && pair_kernel_) {
    if (!ev_res_local_) CheckHip(prost_hip_event_create(&ev_res_local_), "event_create");
    CheckHip(prost_hip_event_record(ev_res_local_, s), "event_record");
  }
  residuals_pending_ = true;
  if (opts_.stepsize_variant == kPDHGStepsResidualGoldstein || opts_.stepsize_variant == kPDHGStepsResidualBoyd) ResolveResiduals();
}

template <typename T>
bool BackendPDHG<T>::CanSpeculate() const {
  // (what the host is about to wait for: the event behind the residual launch, or -- with a communicator -- the one behind the
  // all-reduce on the side stream)
  if (!opts_.allow_speculation || !pair_kernel_ || owned_x1_ != 0 || !(resolve_on_side_ || (ev_res_local_ && !this->comm_)) || spec_valid_) return false;
  if (opts_.stepsize_variant != kPDHGStepsAlg1 && opts_.stepsize_variant != kPDHGStepsAlg2) return false;
  const size_t k = iteration_;
  if (group_max_ >= 2) {
    // a PLAIN group must be what PerformIterations would launch next with a large budget (a smaller budget drops the speculation)
    bool res = false;
    const int g = GroupSize(k, kGroupMax, res);
    if (g < 2 || res) return false;
    return x_spare_.size() == x_.size() && y_spare_.size() == y_.size();
  }
  // a PLAIN pair must be what PerformIterations(budget >= 2) would launch next: no residual sums, no stored intermediate iterate
  if (k < 2 || is_residual_iteration(k) || is_residual_iteration(k + 1) || is_residual_iteration(k + 2)) return false;
  return x_spare_.size() == x_.size() && y_spare_.size() == y_.size();
}

template <typename T>
void BackendPDHG<T>::Speculate() {
  double tau[kGroupMax], sigma[kGroupMax], theta[kGroupMax];
  const T t0 = tau_, s0 = sigma_, th0 = theta_;
  bool res = false;
  const int g = group_max_ >= 2 ? GroupSize(iteration_, kGroupMax, res) : 2;
  spec_count_ = g;
  for (int i = 0; i <= g; i++) {
    spec_tau_[i] = tau_; spec_sigma_[i] = sigma_; spec_theta_[i] = theta_;
    if (i < g && opts_.stepsize_variant == kPDHGStepsAlg2) UpdateAlg2();
  }
  tau_ = t0; sigma_ = s0; theta_ = th0;                  // nothing observable changes until the results are adopted
  for (int i = 0; i < g; i++) { tau[i] = (double)spec_tau_[i]; sigma[i] = (double)spec_sigma_[i]; theta[i] = (double)spec_theta_[i]; }
  if (group_max_ >= 2)
    TimedLaunch(kKernelGroup2 + (g - 2), [&] {
      CheckHip(Api<T>::fused_iterationk(&desc_pair_, g, x_spare_.data(), y_spare_.data(), x_.data(), y_.data(), tau, sigma, theta, 0, nullptr, nullptr,
                                        CurrentStream()), "fused_iterationk");
    });
  else
  TimedLaunch(kKernelPair, [&] {
    CheckHip(Api<T>::fused_iteration2(&desc_pair_, x_spare_.data(), y_spare_.data(), x_.data(), y_.data(), nullptr, nullptr, tau, sigma, theta, 0,
                                      nullptr, nullptr, CurrentStream()), "fused_iteration2");
  });
  spec_iteration_ = iteration_;
  spec_valid_ = true;
  spec_launched_++;            // (KernelTimes counts this launch under the pair kernel whether or not it is adopted: it ran; pair_launches_
                               //  counts adopted ones; solver_state reports speculative_launches / speculative_adopted)
}

template <typename T>
void BackendPDHG<T>::ResolveResiduals() {
  if (!residuals_pending_) return;
  residuals_pending_ = false;
  if (resolve_on_side_) {
    // (with a communicator the sums are being all-reduced on the side stream: the iteration stream is free for the next pair meanwhile)
    if (CanSpeculate()) Speculate();
    CheckHip(prost_hip_event_synchronize(ev_res_done_), "event_synchronize");
    resolve_on_side_ = false;
  }
  else if (CanSpeculate()) {
    Speculate();                                         // the device goes on with the next pair while the host looks at the sums
    CheckHip(prost_hip_event_synchronize(ev_res_local_), "event_synchronize");
  }
  else CheckHip(prost_hip_stream_synchronize(CurrentStream()), "stream_synchronize");
  CheckHip(prost_hip_check_last_error(), "PDHG iteration");
  // the reference reduces in T and takes std::sqrt of the T sums (:433-436)
  this->primal_residual_ = std::sqrt((T)res_host_[0]);
  this->primal_var_norm_ = std::sqrt((T)res_host_[1]);
  this->dual_residual_ = std::sqrt((T)res_host_[2]);
  this->dual_var_norm_ = std::sqrt((T)res_host_[3]);

  const T eps_primal = this->eps_primal(), eps_dual = this->eps_dual();
  switch (opts_.stepsize_variant) {
    case kPDHGStepsResidualGoldstein: {                       // :443-460 (both branches may fire)
      const T scale = eps_dual / eps_primal;
      if (this->dual_residual_ > (scale * this->primal_residual_ * opts_.arg_delta)) {
        tau_ = tau_ / (1 - arg_alpha_);
        sigma_ = sigma_ * (1 - arg_alpha_);
        arg_alpha_ = arg_alpha_ * opts_.arg_nu;
      }
      if (this->dual_residual_ < (scale * this->primal_residual_ / opts_.arg_delta)) {
        tau_ = tau_ * (1 - arg_alpha_);
        sigma_ = sigma_ / (1 - arg_alpha_);
        arg_alpha_ = arg_alpha_ * opts_.arg_nu;
      }
    } break;
    case kPDHGStepsResidualBoyd:                              // :462-476
      if ((this->dual_residual_ < eps_dual) && (opts_.arb_tau * iteration_ > arb_l_)) {
        tau_ /= opts_.arb_delta;
        sigma_ *= opts_.arb_delta;
        arb_u_ = (int)iteration_;
      } else if ((this->primal_residual_ < eps_primal) && (opts_.arb_tau * iteration_ > arb_u_)) {
        tau_ *= opts_.arb_delta;
        sigma_ /= opts_.arb_delta;
        arb_l_ = (int)iteration_;
      }
      break;
    default: break;
  }
}

template <typename T>
void BackendPDHG<T>::UpdateAlg2() {                           // :483-488, computed in double
  theta_ = (T)(1. / std::sqrt(1. + 2. * (double)opts_.alg2_gamma * (double)tau_));
  tau_ = theta_ * tau_;
  sigma_ = sigma_ / theta_;
}

template <typename T>
void BackendPDHG<T>::current_solution(std::vector<T>& primal, std::vector<T>& dual) {
  if (failed_) throw Exception("BackendPDHG: an earlier batch of device-resident iterations failed half-way; the state on the device is undefined. Create a new solver.");
  x_.copy_to(primal);
  y_.copy_to(dual);
}

template <typename T>
void BackendPDHG<T>::ConstraintVariables() {
  if (failed_) throw Exception("BackendPDHG: an earlier batch of device-resident iterations failed half-way; the state on the device is undefined. Create a new solver.");
  void* s = CurrentStream();
  const size_t n = this->problem_->ncols(), m = this->problem_->nrows();
  const device_vector<T>& Tr = this->problem_->scaling_right();
  const device_vector<T>& Sl = this->problem_->scaling_left();
  if (sol_w_.size() != n) sol_w_.resize(n);
  if (sol_z_.size() != m) sol_z_.resize(m);
  if (fused_) {
    RebuildPrevious();
    // rebuild the operator products the generic path keeps resident (callback iterations only)
    device_vector<T> ktyp(n), kx(m), kxp(m);
    if (iteration_ >= 2) this->problem_->linop()->EvalAdjoint(ktyp, y_prev_);      // kty_prev_ = K^T y^(k): zero vector until k = 2
    this->problem_->linop()->Eval(kx, x_);
    if (iteration_ >= 2) this->problem_->linop()->Eval(kxp, x_prev_);              // kx_prev_ is the zero vector after the first iteration
    CheckHip(Api<T>::pdhg_w_variable(sol_w_.data(), x_prev_.data(), x_.data(), Tr.data(), ktyp.data(), (double)tau_, n, s), "w_variable");
    CheckHip(Api<T>::pdhg_z_variable(sol_z_.data(), y_prev_.data(), y_.data(), Sl.data(), kx.data(), kxp.data(), (double)sigma_, (double)theta_, m, s), "z_variable");
    CheckHip(prost_hip_stream_synchronize(s), "stream_synchronize");             // the temporaries above go out of scope
  } else if (op_fused_) {
    // the operator products are not resident: K^T y^k is what the last primal launch stored (kty_; the zero vector after iteration 0, as
    // the reference's kty_prev_), K x and K x_prev are formed now (kx_prev_: the zero vector after the first iteration)
    this->problem_->linop()->Eval(kx_, x_);
    if (iteration_ >= 2) this->problem_->linop()->Eval(kx_prev_, x_prev_);
    else CheckHip(prost_hip_memset(kx_prev_.data(), 0, m * sizeof(T), s), "memset");
    CheckHip(Api<T>::pdhg_w_variable(sol_w_.data(), x_prev_.data(), x_.data(), Tr.data(), kty_.data(), (double)tau_, n, s), "w_variable");
    CheckHip(Api<T>::pdhg_z_variable(sol_z_.data(), y_prev_.data(), y_.data(), Sl.data(), kx_.data(), kx_prev_.data(), (double)sigma_, (double)theta_, m, s), "z_variable");
  } else {
    CheckHip(Api<T>::pdhg_w_variable(sol_w_.data(), x_prev_.data(), x_.data(), Tr.data(), kty_prev_.data(), (double)tau_, n, s), "w_variable");   // :147-160
    CheckHip(Api<T>::pdhg_z_variable(sol_z_.data(), y_prev_.data(), y_.data(), Sl.data(), kx_.data(), kx_prev_.data(), (double)sigma_, (double)theta_, m, s), "z_variable");   // :169-186
  }
}

template <typename T>
void BackendPDHG<T>::current_solution(std::vector<T>& primal_x, std::vector<T>& primal_z, std::vector<T>& dual_y, std::vector<T>& dual_w) {
  x_.copy_to(primal_x);
  y_.copy_to(dual_y);
  ConstraintVariables();
  sol_w_.copy_to(dual_w);
  sol_z_.copy_to(primal_z);
}

template <typename T>
bool BackendPDHG<T>::current_solution_device(const T*& primal_x, const T*& primal_z, const T*& dual_y, const T*& dual_w) {
  ConstraintVariables();
  primal_x = x_.data(); primal_z = sol_z_.data(); dual_y = y_.data(); dual_w = sol_w_.data();
  return true;
}

/// What is RESIDENT after Initialize(): the iterate buffers, the third (spare) buffers of the kernels that ping-pong through
/// them, and the merged mask / b stream of the inpainting shape.  Not resident until somebody reads the solution or the previous
/// iterate (current_solution with z / w, callbacks, solver_compare, the end of the run): z and w (n + m values, kept while small),
/// the three operator products they are formed from (n + 2 m, temporary) and -- on the pair paths that do not keep third buffers
/// (3-D, multi-channel) -- the n + m values RebuildPrevious needs: at most 3 n + 4 m values on top of this figure, ~4.3 GB at
/// 2048 x 2048 x 64 fp32 against 8.6 GB resident.
template <typename T>
size_t BackendPDHG<T>::gpu_mem_amount() const {
  const size_t m = this->problem_->nrows(), n = this->problem_->ncols();
  if (x_.size() == n && n > 0)          // after Initialize(): what the vectors really hold
    return (x_.size() + x_prev_.size() + x_spare_.size() + y_.size() + y_prev_.size() + y_spare_.size() + b_masked_.size() + merged_g_[0].size() + merged_g_[1].size() +
            merged_g_[2].size() + merged_g_[3].size() + merged_g_[4].size() + merged_g_[5].size() + merged_g_[6].size() + kty_.size() + kty_prev_.size() +
            kx_.size() + kx_prev_.size() + temp_.size() + sol_z_.size() + sol_w_.size()) * sizeof(T);
  if (fused_) return 2 * (n + m) * sizeof(T);
  return (4 * (n + m) + std::max(n, m)) * sizeof(T);           // backend_pdhg.cu:504-511
}

template <typename T>
void BackendPDHG<T>::KernelTimes(std::vector<typename Backend<T>::KernelTime>& out) {
  out.clear();
  if (samples_.empty()) return;
  CheckHip(prost_hip_event_synchronize(ev_[ev_used_ - 1]), "event_synchronize");
  double sum[kKernelKinds] = {0};
  size_t cnt[kKernelKinds] = {0};
  for (const Sample& sm : samples_) {
    if (sm.end == kNoEvent) continue;
    float ms = 0;
    CheckHip(prost_hip_event_elapsed_ms(ev_[sm.start], ev_[sm.end], &ms), "event_elapsed");
    sum[sm.kind] += ms; cnt[sm.kind]++;
  }
  const bool d3 = desc_.is3d != 0;
  const char* names[kKernelKinds] = {d3 ? "fused_primal3d_kernel" : "fused_primal2d_kernel", d3 ? "fused_dual3d_kernel" : "fused_dual2d_kernel",
                                     d3 ? "fused_iter3d_kernel" : single_mc_ ? "fused_iter2d_mc_kernel" : "fused_iter2d_kernel",
                                     d3 ? "fused_iter3d_kernel+residuals" : "fused_iter2d_kernel+residuals",
                                     d3 ? "fused_iter3d_x2_kernel" : pair_mc_ ? "fused_iter2d_mc_x2_kernel" : "fused_iter2d_x2_kernel", "fused_iter2d_x2_kernel+mid",
                                     d3 ? "fused_iter3d_x2_kernel+residuals" : pair_mc_ ? "fused_iter2d_mc_x2_kernel+residuals" : "fused_iter2d_x2_kernel+residuals",
                                     "fused_iter2d_x2_kernel+mid+residuals",
                                     "fused_iter2d_xk_kernel<2>", "fused_iter2d_xk_kernel<3>", "fused_iter2d_xk_kernel<4>",
                                     "fused_iter2d_xk_kernel<2>+residuals", "fused_iter2d_xk_kernel<3>+residuals", "fused_iter2d_xk_kernel<4>+residuals"};
  const int iters[kKernelKinds] = {0, 0, 1, 1, 2, 2, 2, 2, 2, 3, 4, 2, 3, 4};
  for (int k = 0; k < kKernelKinds; k++) {
    if (!cnt[k]) continue;
    const int cols = k >= kKernelGroup2 ? prost_hip_fused_iterationk_chunk_cols(&desc_pair_, dtype_id<T>(), iters[k], k >= kKernelGroup2Res)
                     : k >= kKernelPair && pair_kernel_ ? prost_hip_fused_iteration2_chunk_cols(&desc_pair_, dtype_id<T>(), k == kKernelPairRes || k == kKernelPairMidRes)
                     : (k == kKernelPair || k == kKernelPairRes) && pair3d_ ? prost_hip_fused_iteration3d_x2_chunk_cols(&desc_, dtype_id<T>(), k == kKernelPairRes)
                     : (k == kKernelPair || k == kKernelPairRes) && pair_mc_ ? prost_hip_fused_iteration_mc_x2_chunk_cols(&desc_pair_, dtype_id<T>(), k == kKernelPairRes) : 0;
    out.push_back({names[k], sum[k] / cnt[k], cnt[k], launches_[k], iters[k], cols});
  }
  samples_.clear(); ev_used_ = 0; last_end_ = kNoEvent;
  for (int k = 0; k < kKernelKinds; k++) launches_[k] = 0;
}

template class BackendPDHG<float>;
template class BackendPDHG<double>;

}  // namespace prost
