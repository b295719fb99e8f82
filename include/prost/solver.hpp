// prost/solver.hpp -- outer host loop (reference include/prost/solver.hpp:36-113, src/solver.cu).
#ifndef PROST_SOLVER_HPP_
#define PROST_SOLVER_HPP_
#include "prost/common.hpp"

namespace prost {

template <typename T> class Problem;
template <typename T> class Backend;

template <typename T>
class Solver {
 public:
  struct Options {
    T tol_rel_primal, tol_rel_dual, tol_abs_primal, tol_abs_dual;
    int max_iters;
    int num_cback_calls;       ///< how often the intermediate-solution callback fires
    bool verbose;
    std::vector<T> x0, y0;     ///< warm start (PDHG only; ADMM ignores them like the reference)
    bool solve_dual_problem;
    Options() : tol_rel_primal(1e-4), tol_rel_dual(1e-4), tol_abs_primal(1e-4), tol_abs_dual(1e-4), max_iters(1000),
                num_cback_calls(10), verbose(false), solve_dual_problem(false) {}
  };
  enum ConvergenceResult { kConverged, kStoppedMaxIters, kStoppedUser };

  /// (iteration, primal, dual) -> true = converged
  typedef std::function<bool(int, const std::vector<T>&, const std::vector<T>&)> IntermCallback;
  /// polled every iteration; true = stop (user interrupt)
  typedef std::function<bool()> StoppingCallback;

  Solver(shared_ptr<Problem<T>> problem, shared_ptr<Backend<T>> backend);
  virtual ~Solver() {}

  void Initialize();                          // solver.cu:68-120
  ConvergenceResult Solve();                  // solver.cu:123-209
  void Release();
  /// `iters` backend iterations without convergence tests / callbacks (benchmark + test entry)
  void Iterate(int iters);
  /// The loop of Solve() -- backend iterations with the convergence test of solver.cu:141-150 after every observable
  /// iteration -- for exactly `iters` iterations or until the test fires; no callbacks, no read-out of the solution.
  /// What a caller of prost.solve pays per iteration (benchmark entry).  Returns true iff converged.
  bool IterateChecked(int iters);
  /// refreshes cur_*_sol from the device (Backend::current_solution)
  void FetchSolution();
  int iterations_done() const { return iterations_done_; }

  void SetOptions(const Options& opts) { opts_ = opts; }
  const Options& options() const { return opts_; }
  void SetStoppingCallback(const StoppingCallback& cb) { stopping_cb_ = cb; }
  void SetIntermCallback(const IntermCallback& cb) { interm_cb_ = cb; }
  /// Called by Solve() INSTEAD of the read-out into cur_*_sol at the observation that ends the run (convergence, stop, last
  /// iteration) when no intermediate-solution callback is installed: the caller takes the result straight from the backend
  /// (Backend::current_solution_device) while the problem is still in the state it was solved in.  Returns false to decline
  /// (the usual read-out follows).
  typedef std::function<bool()> FinalReadout;
  void SetFinalReadout(const FinalReadout& f) { final_readout_ = f; }

  const std::vector<T>& cur_primal_sol() const;          // x
  const std::vector<T>& cur_dual_sol() const;            // y
  const std::vector<T>& cur_primal_constr_sol() const;   // z
  const std::vector<T>& cur_dual_constr_sol() const;     // w
  shared_ptr<Backend<T>> backend() const { return backend_; }
  shared_ptr<Problem<T>> problem() const { return problem_; }

 protected:
  Options opts_;
  shared_ptr<Problem<T>> problem_;
  shared_ptr<Backend<T>> backend_;
  std::vector<T> cur_primal_sol_, cur_dual_sol_, cur_primal_constr_sol_, cur_dual_constr_sol_;
  IntermCallback interm_cb_;
  StoppingCallback stopping_cb_;
  FinalReadout final_readout_;
  int iterations_done_;
  bool dualized_;
};

}  // namespace prost
#endif
