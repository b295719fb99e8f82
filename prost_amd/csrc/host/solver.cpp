// solver.cpp -- outer host loop: initialisation, iteration, stopping test, callback schedule
// (behaviour of the reference's src/solver.cu).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <list>

#include "prost/backend/backend.hpp"
#include "prost/problem.hpp"
#include "prost/solver.hpp"
#include "prost_hip.h"

namespace prost {

template <typename T>
Solver<T>::Solver(shared_ptr<Problem<T>> problem, shared_ptr<Backend<T>> backend)
    : problem_(problem), backend_(backend), iterations_done_(0), dualized_(false) {}

template <typename T>
void Solver<T>::Initialize() {
  try {
    problem_->Initialize();
  } catch (Exception& e) {
    throw Exception(std::string("Failed to initialize the problem. Reason: ") + e.what());
  }
  if (opts_.solve_dual_problem) {
    problem_->Dualize();
    backend_->SwapGlobalSizes();
    opts_.x0.swap(opts_.y0);
    dualized_ = true;
  }
  try {
    backend_->SetProblem(problem_);
    backend_->SetOptions(opts_);
    StageTimer timer("Backend::Initialize");
    backend_->Initialize();
  } catch (Exception& e) {
    throw Exception(std::string("Failed to initialize the backend. Reason: ") + e.what());
  }
  if (opts_.verbose) {
    const size_t mem = problem_->gpu_mem_amount() + backend_->gpu_mem_amount();
    size_t mem_avail = 0, mem_total = 0;
    prost_hip_mem_info(&mem_avail, &mem_total);
    std::cout << "# primal variables: " << problem_->ncols() << std::endl;
    std::cout << "# dual variables: " << problem_->nrows() << std::endl;
    std::cout << "Memory requirements: " << mem / (1024 * 1024) << "MB (" << mem_avail / (1024 * 1024) << "/"
              << mem_total / (1024 * 1024) << "MB available). Path: " << backend_->path() << "." << std::endl;
  }
  // the host copies of the solution (solver.cu:110-113 sizes them here) are sized by the first FetchSolution instead: at
  // 10^9 entries zero-filling 2 (m + n) host values is more than a second of page faults before the first iteration
  cur_primal_sol_.clear(); cur_primal_constr_sol_.clear(); cur_dual_sol_.clear(); cur_dual_constr_sol_.clear();
  iterations_done_ = 0;
}

template <typename T>
void Solver<T>::Iterate(int iters) {
  // the state is observable after the last iteration only: the backend may fuse the ones before it
  backend_->SetStopOnConvergence(false);
  for (int i = 0; i < iters;) i += backend_->PerformIterations(iters - i);
  iterations_done_ += iters;
}

template <typename T>
bool Solver<T>::IterateChecked(int iters) {
  backend_->SetStopOnConvergence(true);
  for (int i = 0; i < iters;) {
    const int done = backend_->PerformIterations(iters - i);
    i += done;
    iterations_done_ += done;
    // residual accessors wait for the sums of a residual iteration (the only host synchronisation of the loop)
    const T primal_res = backend_->primal_residual(), dual_res = backend_->dual_residual();
    if ((primal_res < backend_->eps_primal()) && (dual_res < backend_->eps_dual())) return true;
  }
  return false;
}

template <typename T>
void Solver<T>::FetchSolution() {
  backend_->current_solution(cur_primal_sol_, cur_primal_constr_sol_, cur_dual_sol_, cur_dual_constr_sol_);
}

template <typename T>
typename Solver<T>::ConvergenceResult Solver<T>::Solve() {
  ConvergenceResult result = kStoppedMaxIters;

  // iterations at which the intermediate-solution callback fires: num_cback_calls points from 0
  // to max_iters-1 (+ the trailing end point linspace appends); fewer than two calls -> only the
  // convergence / last-iteration triggers remain (solver.cu:128-135)
  std::list<double> cb_iters;
  if (opts_.num_cback_calls >= 2) cb_iters = linspace(0, opts_.max_iters - 1, opts_.num_cback_calls);
  else cb_iters.push_back(1e8);

  backend_->SetStopOnConvergence(true);
  for (int i = 0; i < opts_.max_iters; i++) {
    // The loop body below looks at the backend after EVERY iteration in the reference (solver.cu:137-
    // 196), but only three things can change its outcome: the residuals (which only residual
    // iterations update -- the backend never fuses those), the callback schedule and the last
    // iteration (both known here), and a user stop callback (then nothing is fused).  `budget` =
    // iterations up to and including the next scheduled observation (at most 2 with a stopping callback); the backend runs
    // k <= budget of them and the checks run once for the last of these k, whose index is i afterwards.
    const int next_observed = std::min((int)std::ceil(cb_iters.front() < 1e8 ? cb_iters.front() : 1e8), opts_.max_iters - 1);
    // A stopping callback (the MEX gateway's Ctrl-C poll) is asked once per LAUNCH, i.e. after every iteration or every second
    // one: a stop request is honoured at most one iteration later than in the reference, which polls after each (solver.cu:151).
    const int budget = std::max(1, std::min(stopping_cb_ ? 2 : opts_.max_iters, next_observed - i + 1));
    const int done = backend_->PerformIterations(budget);
    i += done - 1;
    iterations_done_ += done;

    const T primal_res = backend_->primal_residual(), dual_res = backend_->dual_residual();
    const T eps_pri = backend_->eps_primal(), eps_dua = backend_->eps_dual();
    bool is_converged = (primal_res < eps_pri) && (dual_res < eps_dua);
    const bool is_stopped = stopping_cb_ ? stopping_cb_() : false;

    if (i >= cb_iters.front() || is_converged || is_stopped || i == (opts_.max_iters - 1)) {
      // the observation that ends the run, nobody to show intermediate solutions to: the caller may take the result directly
      const bool last = is_converged || is_stopped || i == (opts_.max_iters - 1);
      // scheduled observations without a callback only print the residual line: the solution vectors are read when somebody
      // can see them (an installed callback, or the end of the run) -- not 10 times by default (options.m: num_cback_calls = 10)
      if (interm_cb_ || last) {
        if (!(last && !interm_cb_ && final_readout_ && final_readout_())) FetchSolution();
      }
      if (opts_.num_cback_calls >= 1) {
        if (opts_.verbose) {
          const int digits = (int)std::floor(std::log10((double)opts_.max_iters)) + 1;
          std::cout << "It " << std::setw(digits) << (i + 1) << ": " << std::scientific;
          std::cout << "Feas_p=" << std::setprecision(2) << primal_res;
          std::cout << ", Eps_p=" << std::setprecision(2) << eps_pri;
          std::cout << ", Feas_d=" << std::setprecision(2) << dual_res;
          std::cout << ", Eps_d=" << std::setprecision(2) << eps_dua << "; ";
          if (!interm_cb_) std::cout << std::endl;
        }
        if (interm_cb_) {
          if (opts_.solve_dual_problem) is_converged |= interm_cb_(i + 1, cur_dual_sol_, cur_primal_sol_);
          else is_converged |= interm_cb_(i + 1, cur_primal_sol_, cur_dual_sol_);
        }
      }
      cb_iters.pop_front();
    }
    if (is_stopped) {
      if (opts_.verbose) std::cout << "Stopped by user." << std::endl;
      result = kStoppedUser;
      break;
    }
    if (is_converged) {
      if (opts_.verbose) std::cout << "Reached convergence tolerance." << std::endl;
      result = kConverged;
      break;
    }
  }
  if (opts_.solve_dual_problem && dualized_) {        // restore the original problem (solver.cu:198-203)
    problem_->Dualize();
    backend_->SwapGlobalSizes();
    opts_.x0.swap(opts_.y0);
    dualized_ = false;
  }
  if (opts_.verbose && result == kStoppedMaxIters)
    std::cout << "Reached maximum of " << opts_.max_iters << " iterations." << std::endl;
  return result;
}

template <typename T>
void Solver<T>::Release() {
  problem_->Release();
  backend_->Release();
}

// under solve_dual the roles of the solution vectors are exchanged (solver.cu:216-246)
template <typename T> const std::vector<T>& Solver<T>::cur_primal_sol() const { return opts_.solve_dual_problem ? cur_dual_sol_ : cur_primal_sol_; }
template <typename T> const std::vector<T>& Solver<T>::cur_dual_sol() const { return opts_.solve_dual_problem ? cur_primal_sol_ : cur_dual_sol_; }
template <typename T> const std::vector<T>& Solver<T>::cur_primal_constr_sol() const { return opts_.solve_dual_problem ? cur_dual_constr_sol_ : cur_primal_constr_sol_; }
template <typename T> const std::vector<T>& Solver<T>::cur_dual_constr_sol() const { return opts_.solve_dual_problem ? cur_primal_constr_sol_ : cur_dual_constr_sol_; }

template class Solver<float>;
template class Solver<double>;

}  // namespace prost
