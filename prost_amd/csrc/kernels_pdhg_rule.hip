// kernels_pdhg_rule.hip -- the residual-driven step-size rules and the stopping test of PDHG, evaluated ON THE DEVICE.
//
// Reference: BackendPDHG::UpdateResidualsAndStepsizes (backend_pdhg.cu:433-476) takes the square roots of the four residual
// sums, forms eps_primal / eps_dual (backend.hpp:71-74) and adapts tau / sigma by Goldstein's (:443-460) or Boyd's (:462-476)
// rule -- on the host, behind two blocking device-to-host copies; Solver::Solve then tests for convergence (solver.cu:141-150).
// With the reference's DEFAULT options (pdhg.m:4-14: stepsize = 'boyd', residual_iter = 1) that is a host round trip per
// iteration.  Here a one-thread kernel does the same arithmetic, in the same precision and order as host/backend_pdhg.cpp does
// it (both built with -ffp-contract=off), right behind the reduction of the sums (and behind the all-reduce when there is a
// communicator), and leaves
//   * the parameters of the next iterations in the device record (PdhgRecord<T>::p, fused_common.hpp), which the iteration
//     kernels read through scalar loads,
//   * a stop word that turns every later launch of the batch into a no-op once the stopping test has fired, and
//   * a copy of all scalars in pinned host memory (prost_hip_pdhg_rule_state), which the host reads when it next LOOKS at the
//     solver (end of a batch, callback, read-out) -- not after every residual iteration.
#include "pdhg_rule.hpp"

namespace prost_hip {

template <class T>
__global__ void pdhg_rule_begin_kernel(PdhgRecord<T>* r, prost_hip_pdhg_rule_opts o, FusedArgs<T> a, T tau, T sigma, T theta, T arg_alpha, int arb_l,
                                       int arb_u, int stop_on_convergence, prost_hip_pdhg_rule_state* mirror) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  r->stop = 0; r->stop_on_convergence = stop_on_convergence; r->variant = o.variant;
  r->arb_l = arb_l; r->arb_u = arb_u; r->arg_alpha = arg_alpha;
  r->arg_nu = (T)o.arg_nu; r->arg_delta = (T)o.arg_delta; r->arb_delta = (T)o.arb_delta; r->arb_tau = (T)o.arb_tau;
  r->tol_abs_primal = (T)o.tol_abs_primal; r->tol_abs_dual = (T)o.tol_abs_dual; r->tol_rel_primal = (T)o.tol_rel_primal; r->tol_rel_dual = (T)o.tol_rel_dual;
  r->sqrt_rows = o.sqrt_rows; r->sqrt_cols = o.sqrt_cols;
  for (int k = 0; k < 7; k++) { r->g_val[k] = a.g_val[k]; r->f_val[k] = a.f_val[k]; }
  r->Tval = a.Tval; r->Sval = a.Sval;
  r->varT = a.varT; r->Tcls[0] = a.Tcls[0]; r->Tcls[1] = a.Tcls[1]; r->fmor = a.fmor;
  r->p.ec[0] = EdgeTerms<T>(); r->p.ec[1] = EdgeTerms<T>();
  r->evaluations = 0; r->stop_iteration = 0;
  rule_fill_params(r, tau, sigma, theta);
  if (mirror) {
    rule_mirror(r, mirror, tau, sigma, theta);
    for (int k = 0; k < 4; k++) mirror->sums[k] = 0;
    mirror->primal_res = mirror->dual_res = mirror->primal_var = mirror->dual_var = mirror->eps_primal = mirror->eps_dual = 0;
    mirror->last_iteration = 0;
  }
}

template <class T>
__global__ void pdhg_rule_apply_kernel(PdhgRecord<T>* r, const double* __restrict__ sums4, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  rule_apply_device(r, sums4, iteration, mirror);
}

template <class T>
static int rule_begin(void* record, const prost_hip_pdhg_rule_opts* o, const prost_hip_fused_desc* d, double tau, double sigma, double theta, double arg_alpha,
                      int arb_l, int arb_u, int stop_on_convergence, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!record || !o || !d) { set_error("pdhg rule: null argument"); return 1; }
  if (o->variant != PROST_PDHG_RULE_NONE && o->variant != PROST_PDHG_RULE_GOLDSTEIN && o->variant != PROST_PDHG_RULE_BOYD) { set_error("pdhg rule: unknown variant"); return 1; }
  for (int k = 0; k < 7; k++) if (d->f_coeff_ptr[k] || (d->g_coeff_ptr[k] && (k == 0 || k == 2 || k == 4))) {
    set_error("pdhg rule: the step-size dependent prox terms need scalar a, c, e"); return 1;
  }
  const FusedArgs<T> a = make_fused_args<T>(d);
  hipLaunchKernelGGL(pdhg_rule_begin_kernel<T>, dim3(1), dim3(1), 0, as_stream(stream), static_cast<PdhgRecord<T>*>(record), *o, a, (T)tau, (T)sigma, (T)theta,
                     (T)arg_alpha, arb_l, arb_u, stop_on_convergence, mirror);
  PH_LAUNCH_END("pdhg rule begin kernel");
}
template <class T>
static int rule_apply(void* record, const double* sums4, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!record || !sums4) { set_error("pdhg rule: null argument"); return 1; }
  hipLaunchKernelGGL(pdhg_rule_apply_kernel<T>, dim3(1), dim3(1), 0, as_stream(stream), static_cast<PdhgRecord<T>*>(record), sums4, iteration, mirror);
  PH_LAUNCH_END("pdhg rule kernel");
}

template <class T>
static void record_view(const void* record, const void** tau, const void** sigma, const void** theta, const int** stop) {
  const PdhgRecord<T>* r = static_cast<const PdhgRecord<T>*>(record);       // (device address: only offsets are formed)
  if (tau) *tau = &r->p.tau;
  if (sigma) *sigma = &r->p.sigma;
  if (theta) *theta = &r->p.theta;
  if (stop) *stop = &r->stop;
}
}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_pdhg_record_view(const void* record, int dtype, const void** tau, const void** sigma, const void** theta, const int** stop) {
  if (!record || (dtype != 0 && dtype != 1)) { set_error("pdhg_record_view: null record or unknown dtype"); return 1; }
  if (dtype == 0) record_view<float>(record, tau, sigma, theta, stop); else record_view<double>(record, tau, sigma, theta, stop);
  return 0;
}
size_t prost_hip_pdhg_rule_record_bytes(void) { return sizeof(PdhgRecord<double>) > sizeof(PdhgRecord<float>) ? sizeof(PdhgRecord<double>) : sizeof(PdhgRecord<float>); }
int prost_hip_pdhg_rule_begin_f32(void* record, const prost_hip_pdhg_rule_opts* o, const prost_hip_fused_desc* d, double tau, double sigma, double theta, double arg_alpha,
                                  int arb_l, int arb_u, int stop_on_convergence, prost_hip_pdhg_rule_state* mirror, void* stream) {
  return rule_begin<float>(record, o, d, tau, sigma, theta, arg_alpha, arb_l, arb_u, stop_on_convergence, mirror, stream);
}
int prost_hip_pdhg_rule_begin_f64(void* record, const prost_hip_pdhg_rule_opts* o, const prost_hip_fused_desc* d, double tau, double sigma, double theta, double arg_alpha,
                                  int arb_l, int arb_u, int stop_on_convergence, prost_hip_pdhg_rule_state* mirror, void* stream) {
  return rule_begin<double>(record, o, d, tau, sigma, theta, arg_alpha, arb_l, arb_u, stop_on_convergence, mirror, stream);
}
int prost_hip_pdhg_rule_apply_f32(void* record, const double* sums4, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  return rule_apply<float>(record, sums4, iteration, mirror, stream);
}
int prost_hip_pdhg_rule_apply_f64(void* record, const double* sums4, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  return rule_apply<double>(record, sums4, iteration, mirror, stream);
}
}  // extern "C"
