// prost/backend/backend_admm.hpp -- graph-projection ADMM with a CGLS inner solve
// (reference backend_admm.hpp:36-112, src/backend/backend_admm.cu, include/prost/cgls.hpp).
#ifndef PROST_BACKEND_ADMM_HPP_
#define PROST_BACKEND_ADMM_HPP_
#include "prost/backend/backend.hpp"
#include "prost_hip.h"

namespace prost {

template <typename T>
class BackendADMM : public Backend<T> {
 public:
  struct Options {
    double rho0;
    double alpha;            ///< over-relaxation
    double cg_tol_pow, cg_tol_min, cg_tol_max;
    int cg_max_iter;
    int residual_iter;
    T arb_delta, arb_tau, arb_gamma;
    bool cg_graph;           ///< replay the CG rounds of a solve from one captured HIP graph (needs device_cg); off by default:
                             ///< measured 10-25 % SLOWER than the direct launches on ROCm 7.2
    bool fused_rounds;       ///< CG rounds of four launches with the operator applied inside them (needs device_cg and an operator of
                             ///< CSR / gradient blocks); false: the staged rounds with LinearOperator::Eval between the stages
    bool pixel_rounds;       ///< CG rounds of TWO launches for operators [D ; gradient2d] with D coupling the channels of one pixel (needs
                             ///< fused_rounds); false: the four-launch rounds
    bool device_cg;          ///< fused passes + CG scalars resident on the device (default); false = the reference's launch sequence, one blocking nrm2 per scalar
    Options() : rho0(1), alpha(1.7), cg_tol_pow(1.3), cg_tol_min(1e-5), cg_tol_max(1e-8), cg_max_iter(10), residual_iter(1),
                arb_delta(1.05), arb_tau(0.8), arb_gamma(1.01), cg_graph(false), fused_rounds(true), pixel_rounds(true), device_cg(true) {}
  };
  explicit BackendADMM(const Options& opts)
      : opts_(opts), scal_dev_(nullptr), scal_host_(nullptr), workspace_(nullptr), cg_state_(nullptr), cg_workspace_(nullptr), cg_done_host_(nullptr) {}
  virtual ~BackendADMM();

  virtual void Initialize();
  virtual void PerformIteration();
  virtual void Release();
  virtual void current_solution(std::vector<T>& primal, std::vector<T>& dual);
  virtual void current_solution(std::vector<T>& primal_x, std::vector<T>& primal_z, std::vector<T>& dual_y, std::vector<T>& dual_w);
  virtual size_t gpu_mem_amount() const;
  /// "admm:pixel-op": CG rounds of two launches (operators [D ; gradient2d], D pixel-diagonal or any sparse block with one row per
  /// pixel, e.g. a warp matrix); "admm:fused-op": CG rounds of four launches with the
  /// operator inside the stage kernels; "admm:generic": staged rounds
  virtual std::string path() const { return pixel_rounds_ ? "admm:pixel-op" : fused_rounds_ ? "admm:fused-op" : "admm:generic"; }
  T rho() const { return rho_; }
  size_t iteration() const { return iteration_; }
  virtual void KernelTimes(std::vector<typename Backend<T>::KernelTime>& out);
  int last_cg_iterations();                 ///< iterations taken by the most recent CGLS solve (reads the device record)

 private:
  /// y := alpha op(Sigma^(1/2) K Tau^(1/2)) x + beta y   (GemvPrecondK, backend_admm.cu:199-272)
  void Gemv(char op, T alpha, const device_vector<T>& x, T beta, device_vector<T>& y);
  double Nrm2(const device_vector<T>& v, size_t n);
  int Cgls(const device_vector<T>& b, device_vector<T>& x, double shift, double tol, int maxit, device_vector<T>& p,
           device_vector<T>& q, device_vector<T>& r, device_vector<T>& s, int& iterations);   // cgls.hpp:222-371
  void CglsDevice(const device_vector<T>& b, device_vector<T>& x, double shift, double tol, int maxit, device_vector<T>& p,
                  device_vector<T>& q, device_vector<T>& r, device_vector<T>& s);
  void DescribeOperator();
  void PerformIterationFused();
  void PerformIterationUnfused();
  void FinishResiduals(double primal_residual, double primal_var_norm, double dual_residual, double dual_var_norm);
  void GetDual(device_vector<T>& out, const device_vector<T>& half, const device_vector<T>& proj, const device_vector<T>& dual,
               const device_vector<T>& scaling, T expo, size_t n);

  Options opts_;
  device_vector<T> x_half_, z_half_, x_proj_, z_proj_, x_dual_, z_dual_, temp1_, temp2_, temp3_, tmp_n_, tmp_m_;
  double* scal_dev_;
  double* scal_host_;
  void* workspace_;
  void* cg_state_;          ///< device record of the CG scalars (prost_hip_cgls_state_bytes)
  void* cg_workspace_;      ///< per-workgroup partial sums of the fused stages (prost_hip_cgls_workspace_bytes)
  int* cg_done_host_;       ///< pinned word the device stores the solve's epoch to when the stopping test fires
  void* cg_stream_ = nullptr;   ///< stream the captured CG rounds are replayed on
  void* cg_graph_ = nullptr;    ///< executable HIP graph of cg_max_iter rounds
  void* cg_ev_[2] = {nullptr, nullptr};
  int cg_epoch_ = 0;
  bool cg_iters_valid_ = true;
  // CG rounds in four launches (prost_hip_cgls_round_*): the operator as a table of CSR / gradient blocks, one scalar record
  // per round; fused_op_.nblocks == 0: the operator has other blocks (plugins, diags, Kronecker, ...) -> staged rounds
  prost_hip_fused_op fused_op_;
  bool fused_rounds_ = false;
  // CG rounds in two launches (prost_hip_cgls_pixel_round_*): the operator as [D ; gradient2d], second buffers for p and r
  prost_hip_pixel_op pixel_op_;
  bool pixel_rounds_ = false;
  device_vector<T> cg_p_alt_, cg_r_alt_;
  int cg_result_index_ = 0;      ///< record that holds the result of the most recent device solve
  // kernel timing (bench roofline figure): the four launches of ONE round of a sampled solve are bracketed by events
  std::vector<void*> ev_;        ///< pool, eight events per sampled round (begin / end of its four kernels)
  size_t ev_used_ = 0, solves_ = 0, rounds_launched_ = 0;
  T rho_, delta_;
  int arb_u_, arb_l_;
  size_t iteration_;
  int last_cg_iters_ = 0;
  std::vector<shared_ptr<Prox<T>>> prox_g_, prox_f_;
};

}  // namespace prost
#endif
