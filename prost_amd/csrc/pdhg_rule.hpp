// pdhg_rule.hpp -- device code of the residual-driven step-size rules (see kernels_pdhg_rule.hip for what it replaces): shared by the
// stand-alone rule kernel (behind an all-reduce) and the epilogue of the residual fold (kernels_fused_iter.hip: fold4_rule_kernel).
#pragma once
#include "fused_common.hpp"

namespace prost_hip {

template <class T>
__device__ __forceinline__ void rule_fill_params(PdhgRecord<T>* r, T tau, T sigma, T theta) {
  r->p.tau = tau; r->p.sigma = sigma; r->p.theta = theta;
  // what run_iter / run_iter2 evaluate on the host for by-value step sizes (kernels_fused_iter.hip, kernels_fused_iter2.hip)
  r->p.ug = make_uniform_prox<T>(r->g_val, tau * r->Tval);
  r->p.uf = make_uniform_prox<T>(r->f_val, dual_prox_step<T>(sigma, r->Sval, r->fmor));
  if (r->varT) { r->p.ec[0] = make_edge_terms<T>(r->g_val, tau * r->Tcls[0]); r->p.ec[1] = make_edge_terms<T>(r->g_val, tau * r->Tcls[1]); }
}

template <class T>
__device__ __forceinline__ void rule_mirror(const PdhgRecord<T>* r, prost_hip_pdhg_rule_state* m, T ptau, T psigma, T ptheta) {
  if (!m) return;
  m->tau = (double)r->p.tau; m->sigma = (double)r->p.sigma; m->theta = (double)r->p.theta;
  m->prev_tau = (double)ptau; m->prev_sigma = (double)psigma; m->prev_theta = (double)ptheta;
  m->arg_alpha = (double)r->arg_alpha; m->arb_l = r->arb_l; m->arb_u = r->arb_u;
  m->evaluations = r->evaluations; m->stopped = r->stop; m->stop_iteration = r->stop_iteration;
}

// sums4: {primal diff^2, primal var^2, dual diff^2, dual var^2} as the reduction (and the all-reduce) left them
template <class T>
__device__ inline void rule_apply_device(PdhgRecord<T>* r, const double* sums4, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror) {
  if (r->stop) return;                                      // the batch stopped at an earlier iteration: the state stays what it was then
  // host/backend_pdhg.cpp::ResolveResiduals: "the reference reduces in T and takes std::sqrt of the T sums (:433-436)"
  const T primal_res = t_sqrt((T)sums4[0]), primal_var = t_sqrt((T)sums4[1]);
  const T dual_res = t_sqrt((T)sums4[2]), dual_var = t_sqrt((T)sums4[3]);
  // backend.hpp:71-74: sqrt(global size) * tol_abs + tol_rel * var_norm, narrowed to T
  const T eps_primal = (T)(r->sqrt_rows * r->tol_abs_primal + r->tol_rel_primal * primal_var);
  const T eps_dual = (T)(r->sqrt_cols * r->tol_abs_dual + r->tol_rel_dual * dual_var);
  T tau = r->p.tau, sigma = r->p.sigma;
  const T ptau = tau, psigma = sigma, theta = r->p.theta;
  switch (r->variant) {
    case PROST_PDHG_RULE_GOLDSTEIN: {                         // :443-460 (both branches may fire)
      const T scale = eps_dual / eps_primal;
      if (dual_res > (scale * primal_res * r->arg_delta)) {
        tau = tau / (1 - r->arg_alpha);
        sigma = sigma * (1 - r->arg_alpha);
        r->arg_alpha = r->arg_alpha * r->arg_nu;
      }
      if (dual_res < (scale * primal_res / r->arg_delta)) {
        tau = tau * (1 - r->arg_alpha);
        sigma = sigma / (1 - r->arg_alpha);
        r->arg_alpha = r->arg_alpha * r->arg_nu;
      }
    } break;
    case PROST_PDHG_RULE_BOYD:                                // :462-476
      if ((dual_res < eps_dual) && (r->arb_tau * (T)iteration > (T)r->arb_l)) {
        tau /= r->arb_delta;
        sigma *= r->arb_delta;
        r->arb_u = (int)iteration;
      } else if ((primal_res < eps_primal) && (r->arb_tau * (T)iteration > (T)r->arb_u)) {
        tau *= r->arb_delta;
        sigma /= r->arb_delta;
        r->arb_l = (int)iteration;
      }
      break;
    default: break;
  }
  r->evaluations++;
  if (tau != ptau || sigma != psigma) rule_fill_params(r, tau, sigma, theta);
  if (r->stop_on_convergence && (primal_res < eps_primal) && (dual_res < eps_dual)) { r->stop = 1; r->stop_iteration = iteration; }
  if (mirror) {
    rule_mirror(r, mirror, ptau, psigma, theta);
    for (int k = 0; k < 4; k++) mirror->sums[k] = sums4[k];
    mirror->primal_res = (double)primal_res; mirror->dual_res = (double)dual_res; mirror->primal_var = (double)primal_var; mirror->dual_var = (double)dual_var;
    mirror->eps_primal = (double)eps_primal; mirror->eps_dual = (double)eps_dual;
    mirror->last_iteration = iteration;
  }
}


}  // namespace prost_hip
