// prost/backend/backend_pdhg.hpp -- preconditioned PDHG (reference backend_pdhg.hpp:36-158,
// src/backend/backend_pdhg.cu).  Two execution paths with identical arithmetic:
//   fused    K is one gradient2d/3d block, prox_g one elem_operation:1d, prox_f* one
//            elem_operation:norm2 over the gradient components, uniform preconditioners ->
//            two kernels per iteration (prost_hip_fused_primal / _dual), 11 floats/pixel (2-D)
//   generic  any blocks / proxes: the reference's sequence on the generic kernels
#ifndef PROST_BACKEND_PDHG_HPP_
#define PROST_BACKEND_PDHG_HPP_
#include <functional>

#include "prost/backend/backend.hpp"
#include "prost_hip.h"

namespace prost {

template <typename T>
class BackendPDHG : public Backend<T> {
 public:
  enum StepsizeVariant { kPDHGStepsAlg1 = 1, kPDHGStepsAlg2, kPDHGStepsResidualGoldstein, kPDHGStepsResidualBoyd };
  struct Options {
    double tau0, sigma0;
    int residual_iter;
    bool scale_steps_operator;
    T alg2_gamma;
    T arg_alpha0, arg_nu, arg_delta;
    T arb_delta, arb_tau;
    StepsizeVariant stepsize_variant;
    bool allow_fused;          ///< MI355X addition: set false to force the generic path
    bool allow_single_kernel;  ///< MI355X addition: one kernel per non-residual iteration (7 instead of 11 floats/pixel)
    bool allow_pair_kernel;    ///< MI355X addition: two iterations per launch where nobody observes the one in between
    bool allow_arg_fusion;     ///< MI355X addition (generic path): proxes form their argument on the fly, no argument pass
    int allow_op_fusion;       ///< MI355X addition (generic path, round 5; default 0 = off): > 0: the proxes also form the operator products K^T y / K x on the
                               ///< fly and add up the residual terms themselves (operators of sparse / gradient blocks; CSR blocks without row patterns: rows
                               ///< of <= 6 entries on average): K x, K^T y never written, 4 launches per iteration instead of 9, bit-identical.  Off by default:
                               ///< once the stand-alone pattern product had the same row walk, the separate products were as fast or faster at every size
                               ///< measured
    int residual_sums_in_prox; ///< MI355X addition (generic path, round 5): the prox launches that form their argument on the fly add up the residual terms
                               ///< of their elements themselves (no separate reduction over eight vectors).  0: never; 1 (default): problems of >= 2^23
                               ///< elements (x and y together: below that the two small reduction launches are cheaper than the sums' tail in every prox
                               ///< launch); 2: always
    bool allow_speculation;    ///< MI355X addition: the next pair launch is enqueued BEFORE the host waits for the residual sums (alg1 / alg2)
    bool allow_device_rules;   ///< MI355X addition: goldstein / boyd and the stopping test evaluated on the device, one host wait per BATCH of iterations
    int arithmetic;            ///< MI355X addition (round 6): PROST_HIP_ARITH_EXACT (0, default): every iterate rounds like the reference's expressions
                               ///< without contraction, bit for bit with the CPU oracle.  PROST_HIP_ARITH_FMAD (1): the fused iteration kernels may
                               ///< contract multiply-adds (what nvcc's default does to the reference's kernels) and divide through fp32 reciprocal
                               ///< instructions; where a K-iterations-per-launch kernel exists for the problem (fp32 gray-value ROF / TV-L1 shapes) up
                               ///< to 4 iterations run per launch.  Iterates within a stated tolerance of the exact ones (tests/test_gpu_fmad.py).
    int group_max;             ///< MI355X addition (round 6): iterations per launch of the K-iteration kernel; 0 = the class's default, 1 = never (pairs)
    Options() : tau0(1), sigma0(1), residual_iter(1), scale_steps_operator(true), alg2_gamma(0), arg_alpha0(0.5),
                arg_nu(0.95), arg_delta(1.5), arb_delta(1.05), arb_tau(0.8), stepsize_variant(kPDHGStepsResidualBoyd),
                allow_fused(true), allow_single_kernel(true), allow_pair_kernel(true), allow_arg_fusion(true), allow_op_fusion(0), residual_sums_in_prox(1), allow_speculation(true),
                allow_device_rules(true), arithmetic(PROST_HIP_ARITH_EXACT), group_max(0) {}
  };

  explicit BackendPDHG(const Options& opts) : opts_(opts), fused_(false), single_kernel_(false), pair_kernel_(false), res_dev_(nullptr),
                                              res_host_(nullptr), workspace_(nullptr), iteration_(0) {}
  bool single_kernel() const { return single_kernel_; }
  virtual ~BackendPDHG();

  virtual void Initialize();
  virtual void PerformIteration();
  virtual int PerformIterations(int budget);
  virtual void SetStopOnConvergence(bool on) { stop_on_convergence_ = on; }
  /// batches of iterations that ran with the step-size rule and the stopping test on the device (diagnostics)
  size_t device_rule_batches() const { return dev_batches_; }
  /// generic path: the operator products are formed inside the prox kernels (IterationGenericOp)
  bool operator_in_prox_kernels() const { return op_fused_; }
  bool residual_sums_in_prox_launches() const { return res_in_prox_; }
  virtual void Release();
  virtual void current_solution(std::vector<T>& primal, std::vector<T>& dual);
  virtual void current_solution(std::vector<T>& primal_x, std::vector<T>& primal_z, std::vector<T>& dual_y, std::vector<T>& dual_w);
  virtual bool current_solution_device(const T*& primal_x, const T*& primal_z, const T*& dual_y, const T*& dual_w);
  virtual size_t gpu_mem_amount() const;
  virtual void KernelTimes(std::vector<typename Backend<T>::KernelTime>& out);
  virtual std::string path() const;

  /// Column-sharded images (one slab of columns + halo columns per GPU): only the image columns
  /// [x0, x1) of this backend's problem count in the residual sums (x1 == 0: all).  Call before
  /// Initialize(); needs the single-kernel path (gradient2d, L <= 2).
  void SetOwnedColumns(size_t x0, size_t x1) { owned_x0_ = x0; owned_x1_ = x1; }
  /// Column-sharded slabs with the residual-driven rules on the device (round 6): `hook` ENQUEUES the halo exchange on the solver's stream
  /// (RCCL send / recv: device-side, no host wait) and is called inside a device-resident batch whenever `period` iterations have run
  /// since the last exchange; no launch of a batch spans an exchange.  Order of what a batch enqueues per residual iteration:
  /// iteration kernel (+ partial sums) -> fold -> all-reduce of the four sums -> rule kernel -> [exchange when due] -> next iteration kernel.
  /// Without a hook the caller exchanges between its PerformIterations calls (solver_iterate_sharded's host loop).
  void SetExchangeHook(std::function<void()> hook, size_t period, size_t since) { exchange_hook_ = std::move(hook); exchange_period_ = period; since_exchange_ = since; }
  void ClearExchangeHook() { exchange_hook_ = nullptr; }
  size_t since_exchange() const { return since_exchange_; }
  /// the residual-driven rule runs on the device for this problem / option set (batches of iterations, one host wait each)
  bool device_rules() const { return dev_rules_ || dev_rules_generic_; }
  /// device pointers of the current iterate, for halo exchange between slabs: x (n), y (m)
  T* x_data() { spec_valid_ = false; return x_.data(); }      // (the caller may write the iterate: a speculative launch from the old one is forgotten)
  T* y_data() { spec_valid_ = false; return y_.data(); }
  bool single_kernel_path() const { return single_kernel_; }
  virtual size_t pair_launches() const { return pair_launches_; }
  size_t speculative_launches() const { return spec_launched_; }
  size_t speculative_adopted() const { return spec_adopted_; }
  /// one kernel per iteration with residual sums restricted to owned columns: gradient2d with L <= 2 or L = 3 / 4 channels
  bool sharded_path() const { return single_kernel_ || single_mc_; }
  size_t fused_channels() const { return fused_ ? desc_.L : 0; }
  /// verification entry (solver_compare / solver_read): device pointers of the current and the previous iterate; a previous
  /// iterate that a pair launch kept in registers is rebuilt first
  void device_iterates(T*& x, T*& y, T*& x_prev, T*& y_prev) {
    if (fused_) RebuildPrevious();
    x = x_.data(); y = y_.data(); x_prev = x_prev_.data(); y_prev = y_prev_.data();
  }

  // residual accessors pick up sums that are still in flight (see FinishResiduals)
  virtual T primal_residual() const { const_cast<BackendPDHG<T>*>(this)->ResolveResiduals(); return this->primal_residual_; }
  virtual T dual_residual() const { const_cast<BackendPDHG<T>*>(this)->ResolveResiduals(); return this->dual_residual_; }
  virtual T primal_var_norm() const { const_cast<BackendPDHG<T>*>(this)->ResolveResiduals(); return this->primal_var_norm_; }
  virtual T dual_var_norm() const { const_cast<BackendPDHG<T>*>(this)->ResolveResiduals(); return this->dual_var_norm_; }

  T tau() const { return tau_; }
  T sigma() const { return sigma_; }
  T theta() const { return theta_; }
  size_t iteration() const { return iteration_; }

 private:
  bool TryFused();
  void IterationFused(bool residual_iteration);
  void IterationGeneric(bool residual_iteration);
  void IterationGenericOp(bool residual_iteration);      ///< the same iteration with the operator inside the prox kernels (op_fused_)
  bool DescribeGenericOperator(bool stencils_only);
  void IterationPair(bool store_mid, bool residuals);   // iterations k and k+1 in one launch (prost_hip_fused_iteration2)
  void IterationPairMc(bool residuals);   // the same for gradient2d with 2-4 channels (prost_hip_fused_iteration_mc_x2): k + 2 is not a residual iteration
  void IterationPair3D(bool residuals);   // the same for gradient3d (prost_hip_fused_iteration3d_x2): k + 2 is not a residual iteration
  /// tolerance-class arithmetic: iterations k .. k+g-1 in one launch (prost_hip_fused_iterationk, 2 <= g <= group_max_); only the last
  /// one may be a residual iteration (its sums are formed in the kernel)
  void IterationGroup(int g, bool residuals);
  /// the launch PerformIterations would make at iteration k with this budget: 0 = no group launch, else its size; `residuals`: its
  /// last iteration is a residual iteration
  int GroupSize(size_t k, int budget, bool& residuals) const;
 public:
  static constexpr int kGroupMax = 4;
  /// the arithmetic class the iteration kernels of this solve run in (PROST_HIP_ARITH_*) and the largest launch group (0: none)
  int arithmetic() const {
    return fused_ && (desc_pair_.arith == PROST_HIP_ARITH_FMAD || desc_.arith == PROST_HIP_ARITH_FMAD) ? PROST_HIP_ARITH_FMAD : PROST_HIP_ARITH_EXACT;
  }
  int group_max() const { return group_max_; }
 private:
  int group_max_ = 0;
  void RebuildPrevious();                 // x_prev_ / y_prev_ := x^(k-1) / y^(k-1) after a pair that did not store them
  // Speculative next launch.  With alg1 / alg2 nothing on the device depends on the residual sums, but Solver::Solve reads them after
  // every residual iteration: the host wait + the latency of the next launch leave the device idle ~15 us per residual iteration
  // (2.8 % at the headline size).  When the sums are asked for, the pair launch that would follow (iterations k, k+1, plain) is
  // enqueued FIRST, into the spare buffers, and the host waits for the residual launch's event only.  If the solver goes on with a
  // budget >= 2 the results are adopted by exchanging buffers (no launch); anything else -- the solver stops, somebody reads or
  // writes the state -- just forgets them: the iterates the solver can observe are never touched by the speculation.
  bool CanSpeculate() const;
  void Speculate();
  void DropSpeculation() { spec_valid_ = false; }
  bool spec_valid_ = false;
  size_t spec_launched_ = 0, spec_adopted_ = 0;          // statistics: speculative pair launches / those whose results were exchanged in
  size_t spec_iteration_ = 0;
  T spec_tau_[kGroupMax + 1] = {0}, spec_sigma_[kGroupMax + 1] = {0}, spec_theta_[kGroupMax + 1] = {0};   // step sizes of iterations k .. k+g-1 and after the launch
  int spec_count_ = 2;                    // iterations of the speculative launch
  void* ev_res_local_ = nullptr;          // recorded right after a residual launch (no communicator): what the host waits for
  bool is_residual_iteration(size_t k) const { return k == 0 || (k % (size_t)opts_.residual_iter) == 0; }   // backend_pdhg.cu:389
  void FinishResiduals();                 // all-reduce + D2H enqueued; resolved at once only for residual-driven step rules
  void ResolveResiduals();                // wait, sqrt, step-size rules (backend_pdhg.cu:433-476)
  void UpdateAlg2();                      // :483-488

  // ---- step-size rule and stopping test on the device (goldstein / boyd on the one-kernel 2-D paths) ------------------------------
  // The reference adapts tau / sigma on the host from residual norms it copies back at every residual iteration (backend_pdhg.cu:
  // 433-476) -- with its default options (pdhg.m:4-14: boyd, residual_iter = 1) once per ITERATION.  Here a BATCH of up to
  // kDeviceBatch iterations is enqueued without looking at the device: behind the reduction of the four sums (and the all-reduce) a
  // one-thread kernel evaluates the rule and the solver's stopping test (kernels_pdhg_rule.hip), the iteration kernels read their
  // step sizes from its device record, and once the test has fired the remaining launches of the batch return at once.  The host
  // waits ONCE, at the end of the batch, adopts the scalars from the pinned mirror and -- if the batch stopped early -- puts the
  // buffer roles back to what they were after the stopping iteration.
  static constexpr int kDeviceBatch = 240;
  bool dev_rules_ = false;                 // this problem / option set runs that way (Initialize)
  bool dev_rules_generic_ = false;         // ... on the GENERIC path: every prox evaluates from an argument source with record-aware kernels
  bool in_device_batch_ = false;
  bool stop_on_convergence_ = false;
  bool batch_last_launch_evaluated_ = false;
  size_t dev_batches_ = 0;
  void* rule_rec_ = nullptr;               // device: PdhgRecord<T>
  prost_hip_pdhg_rule_state* rule_mirror_ = nullptr;   // pinned host: the scalars of the last evaluation, fetched at the end of a batch ...
  prost_hip_pdhg_rule_state* rule_mirror_dev_ = nullptr;   // ... from the device copy the rule kernels write
  struct BatchMark { size_t iteration_after, pair_launches; T *x, *xp, *y, *yp; bool prev_stale; T *kx, *kxp, *kty, *ktyp;   // (kx .. ktyp: generic path)
                     size_t samples = 0, ev_used = 0, launches[14] = {0};             // kernel timing as it stood after this launch (kKernelKinds == 14)
                     int stale_count = 2; bool stale_group = false; };
  std::vector<BatchMark> batch_marks_;     // one per residual iteration of the running batch: the state to return to if it stopped there
  int PerformIterationsDevice(int budget);
  int PerformIterationsInner(int budget);
  std::function<void()> exchange_hook_;   // slabs: enqueues the halo exchange (SetExchangeHook)
  size_t exchange_period_ = 0, since_exchange_ = 0;
  bool failed_ = false;                    // a device-resident batch threw half-way: the iterate on the device is undefined from then on
  void RestoreRoles(const BatchMark& m);

  Options opts_;
  bool from_matrix_ = false;               // the fused path runs on a block that is gradient2d written out as a sparse matrix (TryFused)
  bool fused_, single_kernel_, pair_kernel_;
  bool pair3d_ = false, pair_mc_ = false;
  size_t pair_launches_ = 0;
  prost_hip_fused_desc desc_;
  // the description the double-iteration kernels run on: desc_, or -- for a BINARY per-pixel coefficient a of prox_g (the
  // inpainting mask of example_tv_inpaint.m:23) -- desc_ with a folded into the b stream (b_masked_, prost_hip_mask_merge)
  prost_hip_fused_desc desc_pair_;
  device_vector<T> b_masked_;
  void TryMaskedPairShape();
  // state: fused keeps x, x_prev, y, y_prev only; generic adds kx, kx_prev, kty, kty_prev, temp
  device_vector<T> x_, y_, x_prev_, y_prev_, temp_, kx_, kty_, kx_prev_, kty_prev_;
  device_vector<T> y_spare_;   // third dual buffer: single-kernel residual iterations read y, y_prev and write y_new
  device_vector<T> x_spare_;   // third primal buffer: pair launches that also store the iterate in between
  // after a pair launch that kept x^(k+1), y^(k+1) in registers, x_prev_ / y_prev_ still hold the pair's
  // INPUT x^k, y^k; whoever needs the true previous iterate first re-runs iteration k from them
  bool prev_stale_ = false;
  device_vector<T> sol_z_, sol_w_;        // constraint variables z, w of current_solution (built on demand)
  void ConstraintVariables();             // sol_z_, sol_w_ := z, w of the current iterate (backend_pdhg.cu:147-186)
  bool residuals_pending_ = false;   // four sums enqueued (device -> pinned host), not yet waited for
  size_t owned_x0_ = 0, owned_x1_ = 0;
  T stale_tau_ = 0, stale_sigma_ = 0, stale_theta_ = 0;   // step sizes of that iteration k
  // after a group launch: x_prev_ / y_prev_ hold the group's input x^k, y^k and x_ / y_ = x^(k+g); stale_count_ = g and the step sizes
  // of iterations k+1 .. k+g-2 (iteration k: stale_tau_ ...) -- RebuildPrevious re-runs g - 1 iterations in the same arithmetic
  int stale_count_ = 2;
  bool stale_group_ = false;
  T stale_tau_more_[kGroupMax] = {0}, stale_sigma_more_[kGroupMax] = {0}, stale_theta_more_[kGroupMax] = {0};
  double* res_dev_;        // 4 doubles: primal (diff^2, var^2), dual (diff^2, var^2)
  /// where the reduction kernels put the four sums: the pinned (device-visible) host buffer, or the device
  /// buffer when an RCCL all-reduce has to run on them first
  bool single_mc_ = false;           // gradient2d, 3 / 4 channels: one kernel per non-residual iteration (prost_hip_fused_iteration_mc_*)
  bool single3d_pw_ = false;         // ... with the planes across the wavefronts of a workgroup on non-residual iterations
  bool single3d_ = false;            // gradient3d: one kernel per non-residual iteration (prost_hip_fused_iteration3d_*)
  bool arg_fused_g_ = false, arg_fused_f_ = false;     // every prox of prox_g_ / prox_fstar_ evaluates from an argument source
  bool op_fused_ = false;                              // ... and from the operator sources: the generic iteration runs without K x / K^T y launches
  prost_hip_fused_op gen_op_;                          // the operator as a table of sparse / gradient blocks (op_fused_)
  const T* view_tau_ = nullptr; const T* view_sigma_ = nullptr; const int* view_stop_ = nullptr;   // device addresses inside rule_rec_ (Prox::StepView)
  bool res_in_prox_ = false;                           // separate products: the prox launches add up the residual terms themselves (ARG 5 / 6)
  void* op_workspace_ = nullptr;                       // residual sums of the prox launches: 2 x kOpSumSlots slots of 4 doubles (primal | dual)
  static constexpr unsigned kOpSumSlots = 8192;
  /// workgroups of one residual launch: every workgroup ends with a block-wide fold of its sums, which 2048 workgroups (two rounds of the
  /// machine at 4 wavefronts per SIMD) spread over 4+ element groups per lane at the sizes where it matters; 8192 -> 2048: 4576 -> 4911 it/s at 2048^2
  static constexpr unsigned kOpLaunchSlots = 2048;
  double* res_target();
  // all-reduce of the sums on a side stream (alg1 / alg2 with a communicator): the iteration stream never waits for the other ranks
  void* side_stream_ = nullptr;
  void* ev_res_ready_ = nullptr;
  void* ev_res_done_ = nullptr;
  bool side_inflight_ = false, resolve_on_side_ = false;
  double* res_host_;       // pinned
  void* workspace_;
  T tau_, sigma_, theta_;
  size_t iteration_;
  int arb_l_, arb_u_;
  T arg_alpha_;
  std::vector<shared_ptr<Prox<T>>> prox_g_, prox_fstar_;
  // kernel timing: event pairs around one launch in eight of every kernel kind
  enum KernelKind { kKernelPrimal = 0, kKernelDual, kKernelIter, kKernelIterRes, kKernelPair, kKernelPairMid, kKernelPairRes, kKernelPairMidRes,
                    kKernelGroup2, kKernelGroup3, kKernelGroup4, kKernelGroup2Res, kKernelGroup3Res, kKernelGroup4Res, kKernelKinds };
  bool BeginSample(int kind);
  void EndSample(bool sampled);
  /// a launch that threw between BeginSample and EndSample: the armed event pair is withdrawn (no later launch of this thread
  /// takes it) and the sample that was never recorded is dropped
  void AbortSample(bool sampled);
  /// BeginSample / launch / EndSample with that clean-up on the exception path
  bool DescribeProxG(ProxDesc& out);         // prox_g as one elem_operation:1d over the whole primal variable (pieces merged)
  device_vector<T> merged_g_[7];             // coefficient vectors assembled from the pieces of prox_g
  template <class F> void TimedLaunch(int kind, F&& launch) {
    const bool sampled = BeginSample(kind);
    try { launch(); } catch (...) { AbortSample(sampled); throw; }
    EndSample(sampled);
  }
  static constexpr size_t kNoEvent = ~(size_t)0;
  static constexpr size_t kMaxSamples = 16384;
  struct Sample { int kind; size_t start, end; };   // indices into ev_
  size_t NewEvent();               // records the next event of the pool on the solver's stream
  std::vector<void*> ev_;          // event pool
  size_t ev_used_ = 0;
  size_t last_end_ = kNoEvent;     // end event of the previous launch while nothing else has been enqueued since
  std::vector<Sample> samples_;
  size_t launches_[kKernelKinds] = {0};
};

}  // namespace prost
#endif
