// prost/backend/backend.hpp -- algorithm interface (reference include/prost/backend/backend.hpp:37-95).
#ifndef PROST_BACKEND_BACKEND_HPP_
#define PROST_BACKEND_BACKEND_HPP_
#include <cmath>
#include <string>
#include <vector>

#include "prost/problem.hpp"
#include "prost/solver.hpp"

namespace prost {

template <typename T>
class Backend {
 public:
  Backend() : primal_var_norm_(0), dual_var_norm_(0), primal_residual_(0), dual_residual_(0), comm_(nullptr),
              global_nrows_(0), global_ncols_(0), time_kernels_(false), sample_every_(8) {}
  virtual ~Backend() {}

  void SetProblem(shared_ptr<Problem<T>> problem) { problem_ = problem; }
  void SetOptions(const typename Solver<T>::Options& opts) { solver_opts_ = opts; }

  virtual void Initialize() = 0;
  virtual void PerformIteration() = 0;
  /// MI355X addition: run 1 <= k <= budget iterations and return k.  `budget` counts the iterations
  /// up to AND INCLUDING the next one after which the caller looks at the backend's state (solution
  /// read-out, callback, stopping test, end of the run).  A backend may fuse iterations whose
  /// intermediate state nobody observes; after the call the state must be exactly what k calls of
  /// PerformIteration() leave (including whatever current_solution() needs of iteration k-1).
  /// Residual iterations update the residuals as usual.  Default: one iteration.
  virtual int PerformIterations(int budget) { (void)budget; PerformIteration(); return 1; }
  /// MI355X addition.  true: the caller runs the loop of Solver::Solve -- it tests primal_residual() < eps_primal() &&
  /// dual_residual() < eps_dual() after every PerformIterations call and stops when the test holds (solver.cu:141-150).  A backend
  /// that evaluates the same test on the device may then end a PerformIterations(budget) call EARLY, at the first residual iteration
  /// at which it holds, returning the iterations run up to and including that one.  false (default): every iteration asked for runs.
  virtual void SetStopOnConvergence(bool on) { (void)on; }
  virtual void Release() = 0;

  virtual void current_solution(std::vector<T>& primal_sol, std::vector<T>& dual_sol) = 0;
  virtual void current_solution(std::vector<T>& primal_x, std::vector<T>& primal_z, std::vector<T>& dual_y,
                                std::vector<T>& dual_w) = 0;
  /// The same four vectors left ON THE DEVICE (pointers valid until the next iteration or Release): lets the caller stream a
  /// large result to wherever it is needed instead of passing through four host vectors.  false: not provided, call
  /// current_solution.
  virtual bool current_solution_device(const T*& primal_x, const T*& primal_z, const T*& dual_y, const T*& dual_w) {
    (void)primal_x; (void)primal_z; (void)dual_y; (void)dual_w;
    return false;
  }

  /// launches that ran two iterations at once so far (diagnostics: 0 for backends that never fuse iterations)
  virtual size_t pair_launches() const { return 0; }

  virtual T primal_residual() const { return primal_residual_; }
  virtual T dual_residual() const { return dual_residual_; }
  virtual T primal_var_norm() const { return primal_var_norm_; }
  virtual T dual_var_norm() const { return dual_var_norm_; }
  /// backend.hpp:71-74 ; with a communicator the sizes are the global ones
  virtual T eps_primal() const {
    return std::sqrt((double)(global_nrows_ ? global_nrows_ : problem_->nrows())) * solver_opts_.tol_abs_primal + solver_opts_.tol_rel_primal * primal_var_norm();
  }
  virtual T eps_dual() const {
    return std::sqrt((double)(global_ncols_ ? global_ncols_ : problem_->ncols())) * solver_opts_.tol_abs_dual + solver_opts_.tol_rel_dual * dual_var_norm();
  }
  virtual size_t gpu_mem_amount() const = 0;

  /// multi-GPU batches: the 4 residual sums are all-reduced over `comm` (an RCCL communicator made
  /// by prost_hip_comm_create) so every rank takes identical stopping / step-size decisions.
  void SetCommunicator(void* comm, size_t global_nrows, size_t global_ncols) { comm_ = comm; global_nrows_ = global_nrows; global_ncols_ = global_ncols; }
  /// Problem::Dualize() exchanges rows and columns; the global sizes follow (called by Solver when it dualizes)
  void SwapGlobalSizes() { const size_t t = global_nrows_; global_nrows_ = global_ncols_; global_ncols_ = t; }
  /// record HIP events around a sample of the iteration kernels' launches (bench roofline figure)
  /// `every`: one launch in `every` of each kernel kind is bracketed (1 = all of them: short runs; the markers cost
  /// launch pipelining, so long runs sample one in eight)
  void EnableKernelTiming(bool on, int every = 8) { time_kernels_ = on; sample_every_ = every < 1 ? 1 : every; }
  struct KernelTime {
    std::string name;            ///< kernel symbol as rocprofv3 --kernel-trace reports it
    double avg_ms;               ///< mean launch duration over the sampled launches
    size_t sampled;              ///< launches timed
    size_t launches;             ///< all launches of this kind while timing was enabled
    int iterations_per_launch;   ///< PDHG iterations one launch performs (0: a fraction -- one of two passes)
    int chunk_cols;              ///< image columns per wavefront of this launch geometry (0: not applicable / unknown)
  };
  /// mean milliseconds per launch of every kernel kind sampled since the last call
  virtual void KernelTimes(std::vector<KernelTime>& out) { out.clear(); }
  /// short description of the execution path ("pdhg:fused-grad2d", "pdhg:generic", "admm:fused-op", "admm:generic")
  virtual std::string path() const = 0;

 protected:
  shared_ptr<Problem<T>> problem_;
  typename Solver<T>::Options solver_opts_;
  T primal_var_norm_, dual_var_norm_, primal_residual_, dual_residual_;
  void* comm_;
  size_t global_nrows_, global_ncols_;
  bool time_kernels_;
  int sample_every_;
};

}  // namespace prost
#endif
