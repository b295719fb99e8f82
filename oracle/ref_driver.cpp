// ref_driver.cpp -- thin extern "C" driver around the REAL reference code, compiled from the
// sources where they lie under /root/reference (see Makefile.ref).  TEST INFRASTRUCTURE ONLY:
// used to pin oracle/prost_oracle.cpp and to generate tests/golden/*.npz.
//
// What is the reference's own code here (compiled unmodified, in place):
//   * include/prost/prox/elemop/{function_1d,elem_operation_1d,elem_operation_norm2}.hpp,
//     include/prost/prox/{vector,helper}.hpp           -- all elementwise prox arithmetic
//   * src/backend/backend_pdhg.cu, src/problem.cu, src/linop/{linearoperator,
//     dual_linearoperator,block}.cu, src/prox/{prox,prox_moreau,prox_zero}.cu, src/common.cu
//     -- the PDHG iteration, residuals, step rules, preconditioners, normest, Moreau, csr2csc
//   thrust::device_vector is rocThrust's host (CPP) backend: -DTHRUST_DEVICE_SYSTEM=..._CPP.
// What is NOT buildable here (needs a GPU launch, cuSPARSE, cuBLAS or cudaMemGetInfo):
//   block_gradient2d/3d.cu, block_diags.cu, block_sparse.cu, prox_elem_operation.inl,
//   prox_ind_epi_quad.cu (kernel wrapper), backend_admm.cu, cgls.hpp, solver.cu.
//   Leaf operators are therefore supplied through the reference's own plugin interface
//   (prost::Block / prost::ProxSeparableSum subclasses below): blocks call back into the
//   caller, elementwise proxes run the reference's ELEM_OPERATION functors in a host loop that
//   mirrors the kernel body prox_elem_operation.inl:59-94.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
using std::abs; using std::max; using std::min;

#include "prost/backend/backend_pdhg.hpp"
#include "prost/common.hpp"
#include "prost/exception.hpp"
#include "prost/linop/block.hpp"
#include "prost/linop/linearoperator.hpp"
#include "prost/problem.hpp"
#include "prost/prox/elemop/elem_operation_1d.hpp"
#include "prost/prox/elemop/elem_operation_norm2.hpp"
#include "prost/prox/elemop/function_1d.hpp"
#include "prost/prox/helper.hpp"
#include "prost/prox/prox_moreau.hpp"
#include "prost/prox/prox_separable_sum.hpp"
#include "prost/prox/prox_zero.hpp"

using namespace prost;

static thread_local std::string g_err;
extern "C" const char* ref_last_error() { return g_err.c_str(); }

// ---- elementwise ops: host loop around the reference functor (kernel body restated) ----
template <typename T, class OP>
static void run_elem(T* d_res, const T* d_arg, const T* d_tau, T tau, bool invert_tau, size_t count, size_t dim,
                     bool interleaved, const T* const* cptr, const T* cval) {
  typedef SharedMem<typename OP::SharedMemType, typename OP::GetSharedMemCount> SM;
  alignas(SM) char smbuf[sizeof(SM)];
  SM& sh_mem = *reinterpret_cast<SM*>(smbuf);   // never dereferenced: GetSharedMemCount == 0 for every in-tree op
  for (size_t tx = 0; tx < count; tx++) {
    Vector<T> res(count, dim, interleaved, tx, d_res);
    const Vector<const T> arg(count, dim, interleaved, tx, d_arg);
    const Vector<const T> tau_diag(count, dim, interleaved, tx, d_tau);
    T coeffs_local[OP::kCoeffsCount];
    for (size_t i = 0; i < OP::kCoeffsCount; i++) coeffs_local[i] = cptr[i] == nullptr ? cval[i] : cptr[i][tx];
    OP op(coeffs_local, dim, sh_mem);
    op(res, arg, tau_diag, tau, invert_tau);
  }
}

template <typename T, template <typename, class> class OPT>
static void dispatch_fn(int fn, T* res, const T* arg, const T* tau_d, T tau, bool inv, size_t count, size_t dim,
                        bool il, const T* const* cp, const T* cv) {
#define CASE(id, F) case id: run_elem<T, OPT<T, F<T>>>(res, arg, tau_d, tau, inv, count, dim, il, cp, cv); break;
  switch (fn) {
    CASE(0, Function1DZero) CASE(1, Function1DAbs) CASE(2, Function1DSquare) CASE(3, Function1DIndLeq0)
    CASE(4, Function1DIndGeq0) CASE(5, Function1DIndEq0) CASE(6, Function1DIndBox01) CASE(7, Function1DMaxPos0)
    CASE(8, Function1DL0) CASE(9, Function1DHuber) CASE(10, Function1DLq) CASE(11, Function1DLqPlusEps)
    CASE(12, Function1DTruncLinear) CASE(13, Function1DTruncQuad)
    default: throw Exception("bad function id");
  }
#undef CASE
}

template <typename T>
static void elem_any(int op, int fn, T* res, const T* arg, const T* tau_d, T tau, bool inv, size_t count, size_t dim,
                     bool il, const T* const* cp, const T* cv) {
  if (op == 0) dispatch_fn<T, ElemOperation1D>(fn, res, arg, tau_d, tau, inv, count, 1, il, cp, cv);
  else dispatch_fn<T, ElemOperationNorm2>(fn, res, arg, tau_d, tau, inv, count, dim, il, cp, cv);
}

// plugin: ProxSeparableSum subclass (reference plugin API, prox_separable_sum.hpp:47-86)
template <typename T>
class HostProxElem : public ProxSeparableSum<T> {
 public:
  HostProxElem(int op, int fn, size_t index, size_t count, size_t dim, bool interleaved, bool diagsteps,
               std::array<std::vector<T>, 7> coeffs)
      : ProxSeparableSum<T>(index, count, op == 0 ? 1 : dim, interleaved, diagsteps), op_(op), fn_(fn), coeffs_(coeffs) {}
  virtual size_t gpu_mem_amount() const { return 0; }

 protected:
  virtual void EvalLocal(const typename thrust::device_vector<T>::iterator& result_beg,
                         const typename thrust::device_vector<T>::iterator& result_end,
                         const typename thrust::device_vector<T>::const_iterator& arg_beg,
                         const typename thrust::device_vector<T>::const_iterator& arg_end,
                         const typename thrust::device_vector<T>::const_iterator& tau_beg,
                         const typename thrust::device_vector<T>::const_iterator& tau_end, T tau, bool invert_tau) {
    const T* cp[7]; T cv[7];
    for (int i = 0; i < 7; i++) {
      if (coeffs_[i].size() > 1) { cp[i] = coeffs_[i].data(); cv[i] = 0; } else { cp[i] = nullptr; cv[i] = coeffs_[i][0]; }
    }
    elem_any<T>(op_, fn_, thrust::raw_pointer_cast(&(*result_beg)), thrust::raw_pointer_cast(&(*arg_beg)),
                thrust::raw_pointer_cast(&(*tau_beg)), tau, invert_tau, this->count_, this->dim_, this->interleaved_, cp, cv);
  }
  int op_, fn_;
  std::array<std::vector<T>, 7> coeffs_;
};

// plugin: Block subclass calling back into the caller for K x / K^T y (accumulating)
typedef void (*ref_block_cb)(void* user, int adjoint, void* res, const void* rhs);
template <typename T>
class CbBlock : public Block<T> {
 public:
  CbBlock(size_t row, size_t col, size_t nrows, size_t ncols, ref_block_cb cb, void* user, double rs, double cs)
      : Block<T>(row, col, nrows, ncols), cb_(cb), user_(user), rs_(rs), cs_(cs) {}
  virtual T row_sum(size_t, T) const { return (T)rs_; }
  virtual T col_sum(size_t, T) const { return (T)cs_; }
  virtual size_t gpu_mem_amount() const { return 0; }

 protected:
  virtual void EvalLocalAdd(const typename device_vector<T>::iterator& res_begin, const typename device_vector<T>::iterator&,
                            const typename device_vector<T>::const_iterator& rhs_begin,
                            const typename device_vector<T>::const_iterator&) {
    cb_(user_, 0, thrust::raw_pointer_cast(&(*res_begin)), thrust::raw_pointer_cast(&(*rhs_begin)));
  }
  virtual void EvalAdjointLocalAdd(const typename device_vector<T>::iterator& res_begin,
                                   const typename device_vector<T>::iterator&,
                                   const typename device_vector<T>::const_iterator& rhs_begin,
                                   const typename device_vector<T>::const_iterator&) {
    cb_(user_, 1, thrust::raw_pointer_cast(&(*res_begin)), thrust::raw_pointer_cast(&(*rhs_begin)));
  }
  ref_block_cb cb_; void* user_; double rs_, cs_;
};

struct RefProblem {
  int dtype;
  std::shared_ptr<Problem<float>> pf;
  std::shared_ptr<Problem<double>> pd;
  bool initialized = false;
};

struct ref_pdhg_opts {
  double tau0, sigma0; int residual_iter; int scale_steps_operator;
  double alg2_gamma, arg_alpha0, arg_nu, arg_delta, arb_delta, arb_tau; int stepsize;
};
struct ref_tol_opts { double tol_rel_primal, tol_rel_dual, tol_abs_primal, tol_abs_dual; int solve_dual; };

template <typename T>
static std::shared_ptr<Prox<T>> make_elem(int op, int fn, size_t idx, size_t count, size_t dim, int il, int ds,
                                          const double* const* coeff, const size_t* len, int moreau_depth) {
  std::array<std::vector<T>, 7> c;
  for (int i = 0; i < 7; i++) c[i] = std::vector<T>(coeff[i], coeff[i] + len[i]);
  std::shared_ptr<Prox<T>> p(new HostProxElem<T>(op, fn, idx, count, dim, il, ds, c));
  for (int k = 0; k < moreau_depth; k++) p = std::shared_ptr<Prox<T>>(new ProxMoreau<T>(p));
  return p;
}

template <typename T>
static void run_pdhg(std::shared_ptr<Problem<T>> prob, bool& initialized, const ref_pdhg_opts* po, const ref_tol_opts* to,
                     const double* x0, size_t nx0, const double* y0, size_t ny0, int iters, double* x, double* y,
                     double* z, double* w, double* scal) {
  typename BackendPDHG<T>::Options o;
  o.tau0 = po->tau0; o.sigma0 = po->sigma0; o.residual_iter = po->residual_iter;
  o.scale_steps_operator = po->scale_steps_operator; o.alg2_gamma = (T)po->alg2_gamma;
  o.arg_alpha0 = (T)po->arg_alpha0; o.arg_nu = (T)po->arg_nu; o.arg_delta = (T)po->arg_delta;
  o.arb_delta = (T)po->arb_delta; o.arb_tau = (T)po->arb_tau;
  o.stepsize_variant = (typename BackendPDHG<T>::StepsizeVariant)(po->stepsize + 1);   // enum starts at 1 (backend_pdhg.hpp:44)
  typename Solver<T>::Options so;
  so.tol_rel_primal = (T)to->tol_rel_primal; so.tol_rel_dual = (T)to->tol_rel_dual;
  so.tol_abs_primal = (T)to->tol_abs_primal; so.tol_abs_dual = (T)to->tol_abs_dual;
  so.max_iters = iters; so.num_cback_calls = 0; so.verbose = false; so.solve_dual_problem = to->solve_dual;
  if (x0 && nx0) so.x0 = std::vector<T>(x0, x0 + nx0);
  if (y0 && ny0) so.y0 = std::vector<T>(y0, y0 + ny0);
  // Solver::Initialize (solver.cu:68-90) minus the verbose/cudaMemGetInfo block
  if (!initialized) { prob->Initialize(); initialized = true; }
  if (so.solve_dual_problem) { prob->Dualize(); so.x0.swap(so.y0); }
  std::shared_ptr<BackendPDHG<T>> be(new BackendPDHG<T>(o));
  be->SetProblem(prob);
  be->SetOptions(so);
  be->Initialize();
  for (int i = 0; i < iters; i++) be->PerformIteration();
  std::vector<T> px(prob->ncols()), pz(prob->nrows()), dy(prob->nrows()), dw(prob->ncols());
  be->current_solution(px, pz, dy, dw);
  if (so.solve_dual_problem) { px.swap(dy); pz.swap(dw); }
  if (x) std::copy(px.begin(), px.end(), x);
  if (z) std::copy(pz.begin(), pz.end(), z);
  if (y) std::copy(dy.begin(), dy.end(), y);
  if (w) std::copy(dw.begin(), dw.end(), w);
  scal[0] = be->primal_residual(); scal[1] = be->dual_residual();
  scal[2] = be->primal_var_norm(); scal[3] = be->dual_var_norm();
  scal[4] = be->eps_primal(); scal[5] = be->eps_dual();
  if (so.solve_dual_problem) prob->Dualize();
}

#define REF_TRY try {
#define REF_CATCH } catch (const std::exception& e) { g_err = e.what(); return 1; } return 0;

extern "C" {

int ref_prox_elem(int dtype, int op, int fn, void* res, const void* arg, const void* tau_diag, double tau, int invert,
                  size_t count, size_t dim, int interleaved, const void* const* coeff_ptr, const double* coeff_val) {
  REF_TRY
  if (dtype == 0) {
    const float* cp[7]; float cv[7];
    for (int i = 0; i < 7; i++) { cp[i] = (const float*)(coeff_ptr ? coeff_ptr[i] : nullptr); cv[i] = (float)coeff_val[i]; }
    elem_any<float>(op, fn, (float*)res, (const float*)arg, (const float*)tau_diag, (float)tau, invert, count, dim, interleaved, cp, cv);
  } else {
    const double* cp[7]; double cv[7];
    for (int i = 0; i < 7; i++) { cp[i] = (const double*)(coeff_ptr ? coeff_ptr[i] : nullptr); cv[i] = coeff_val[i]; }
    elem_any<double>(op, fn, (double*)res, (const double*)arg, (const double*)tau_diag, tau, invert, count, dim, interleaved, cp, cv);
  }
  REF_CATCH
}

// helper::ProjectEpiQuadNd on `count` planar (x_1..x_dim, y) points, in place semantics of the
// kernel call site (prox_ind_epi_quad.cu:69): x0 aliases x.
int ref_project_epi_quad(int dtype, void* xy, size_t count, size_t dim, const void* y0, const void* alpha) {
  REF_TRY
  for (size_t tx = 0; tx < count; tx++) {
    if (dtype == 0) {
      float* d = (float*)xy; Vector<float> x(count, dim, false, tx, d);
      helper::ProjectEpiQuadNd<float>(x, ((const float*)y0)[tx], ((const float*)alpha)[tx], x, d[count * dim + tx], dim);
    } else {
      double* d = (double*)xy; Vector<double> x(count, dim, false, tx, d);
      helper::ProjectEpiQuadNd<double>(x, ((const double*)y0)[tx], ((const double*)alpha)[tx], x, d[count * dim + tx], dim);
    }
  }
  REF_CATCH
}

int ref_csr2csc(int dtype, int n, int m, int nz, void* a, int* col_idx, int* row_start, void* csc_a, int* row_idx, int* col_start) {
  REF_TRY
  if (dtype == 0) csr2csc<float>(n, m, nz, (float*)a, col_idx, row_start, (float*)csc_a, row_idx, col_start);
  else csr2csc<double>(n, m, nz, (double*)a, col_idx, row_start, (double*)csc_a, row_idx, col_start);
  REF_CATCH
}
int ref_linspace(double start, double end, int num, double* out) {
  std::list<double> l = linspace<double>(start, end, num);
  size_t k = 0; for (double v : l) out[k++] = v;
  return (int)k;
}

void* ref_problem_create(int dtype, size_t nrows, size_t ncols) {
  RefProblem* p = new RefProblem; p->dtype = dtype;
  if (dtype == 0) { p->pf.reset(new Problem<float>()); p->pf->SetDimensions(nrows, ncols); }
  else { p->pd.reset(new Problem<double>()); p->pd->SetDimensions(nrows, ncols); }
  return p;
}
void ref_problem_destroy(void* h) { delete (RefProblem*)h; }
int ref_problem_add_block_cb(void* h, size_t row, size_t col, size_t nrows, size_t ncols, ref_block_cb cb, void* user,
                             double row_sum, double col_sum) {
  RefProblem* p = (RefProblem*)h;
  REF_TRY
  if (p->dtype == 0) p->pf->AddBlock(std::shared_ptr<Block<float>>(new CbBlock<float>(row, col, nrows, ncols, cb, user, row_sum, col_sum)));
  else p->pd->AddBlock(std::shared_ptr<Block<double>>(new CbBlock<double>(row, col, nrows, ncols, cb, user, row_sum, col_sum)));
  REF_CATCH
}
// which: 0 g, 1 f, 2 gstar, 3 fstar
int ref_problem_add_prox_elem(void* h, int which, int op, int fn, size_t idx, size_t count, size_t dim, int interleaved,
                              int diagsteps, const double* const* coeff, const size_t* len, int moreau_depth) {
  RefProblem* p = (RefProblem*)h;
  REF_TRY
  if (p->dtype == 0) {
    auto q = make_elem<float>(op, fn, idx, count, dim, interleaved, diagsteps, coeff, len, moreau_depth);
    if (which == 0) p->pf->AddProx_g(q); else if (which == 1) p->pf->AddProx_f(q); else if (which == 2) p->pf->AddProx_gstar(q); else p->pf->AddProx_fstar(q);
  } else {
    auto q = make_elem<double>(op, fn, idx, count, dim, interleaved, diagsteps, coeff, len, moreau_depth);
    if (which == 0) p->pd->AddProx_g(q); else if (which == 1) p->pd->AddProx_f(q); else if (which == 2) p->pd->AddProx_gstar(q); else p->pd->AddProx_fstar(q);
  }
  REF_CATCH
}
int ref_problem_add_prox_zero(void* h, int which, size_t idx, size_t size) {
  RefProblem* p = (RefProblem*)h;
  REF_TRY
  if (p->dtype == 0) {
    std::shared_ptr<Prox<float>> q(new ProxZero<float>(idx, size));
    if (which == 0) p->pf->AddProx_g(q); else if (which == 1) p->pf->AddProx_f(q); else if (which == 2) p->pf->AddProx_gstar(q); else p->pf->AddProx_fstar(q);
  } else {
    std::shared_ptr<Prox<double>> q(new ProxZero<double>(idx, size));
    if (which == 0) p->pd->AddProx_g(q); else if (which == 1) p->pd->AddProx_f(q); else if (which == 2) p->pd->AddProx_gstar(q); else p->pd->AddProx_fstar(q);
  }
  REF_CATCH
}
int ref_problem_set_scaling(void* h, int type, double alpha, const double* left, size_t nl, const double* right, size_t nr) {
  RefProblem* p = (RefProblem*)h;
  REF_TRY
  if (p->dtype == 0) {
    if (type == 0) p->pf->SetScalingAlpha((float)alpha); else if (type == 1) p->pf->SetScalingIdentity();
    else p->pf->SetScalingCustom(std::vector<float>(left, left + nl), std::vector<float>(right, right + nr));
  } else {
    if (type == 0) p->pd->SetScalingAlpha(alpha); else if (type == 1) p->pd->SetScalingIdentity();
    else p->pd->SetScalingCustom(std::vector<double>(left, left + nl), std::vector<double>(right, right + nr));
  }
  REF_CATCH
}
int ref_problem_initialize(void* h) {
  RefProblem* p = (RefProblem*)h;
  REF_TRY
  if (!p->initialized) { if (p->dtype == 0) p->pf->Initialize(); else p->pd->Initialize(); p->initialized = true; }
  REF_CATCH
}
int ref_problem_get_scaling(void* h, double* left, double* right) {
  RefProblem* p = (RefProblem*)h;
  REF_TRY
  if (p->dtype == 0) { auto& l = p->pf->scaling_left(); auto& r = p->pf->scaling_right();
    for (size_t i = 0; i < l.size(); i++) left[i] = l[i]; for (size_t i = 0; i < r.size(); i++) right[i] = r[i]; }
  else { auto& l = p->pd->scaling_left(); auto& r = p->pd->scaling_right();
    for (size_t i = 0; i < l.size(); i++) left[i] = l[i]; for (size_t i = 0; i < r.size(); i++) right[i] = r[i]; }
  REF_CATCH
}
int ref_problem_normest(void* h, double* out) {
  RefProblem* p = (RefProblem*)h;
  REF_TRY
  if (p->dtype == 0) *out = p->pf->normest(); else *out = p->pd->normest();
  REF_CATCH
}
// Eval a prox list entry through Prox::Eval(device vectors) (prox.cu:27-43); used for Moreau pins
int ref_prox_elem_eval(int dtype, int op, int fn, size_t count, size_t dim, int interleaved, const double* const* coeff,
                       const size_t* len, int moreau_depth, void* res, const void* arg, const void* tau_diag, double tau) {
  REF_TRY
  if (dtype == 0) {
    auto q = make_elem<float>(op, fn, 0, count, dim, interleaved, true, coeff, len, moreau_depth); q->Initialize();
    size_t n = q->size(); thrust::device_vector<float> r(n), a((const float*)arg, (const float*)arg + n), t((const float*)tau_diag, (const float*)tau_diag + n);
    q->Eval(r, a, t, (float)tau); thrust::copy(r.begin(), r.end(), (float*)res);
  } else {
    auto q = make_elem<double>(op, fn, 0, count, dim, interleaved, true, coeff, len, moreau_depth); q->Initialize();
    size_t n = q->size(); thrust::device_vector<double> r(n), a((const double*)arg, (const double*)arg + n), t((const double*)tau_diag, (const double*)tau_diag + n);
    q->Eval(r, a, t, tau); thrust::copy(r.begin(), r.end(), (double*)res);
  }
  REF_CATCH
}
int ref_pdhg_run(void* h, const ref_pdhg_opts* po, const ref_tol_opts* to, const double* x0, size_t nx0, const double* y0,
                 size_t ny0, int iters, double* x, double* z, double* y, double* w, double* scal6) {
  RefProblem* p = (RefProblem*)h;
  REF_TRY
  if (p->dtype == 0) run_pdhg<float>(p->pf, p->initialized, po, to, x0, nx0, y0, ny0, iters, x, y, z, w, scal6);
  else run_pdhg<double>(p->pd, p->initialized, po, to, x0, nx0, y0, ny0, iters, x, y, z, w, scal6);
  REF_CATCH
}
int ref_rand(void) { return std::rand(); }
void ref_srand(unsigned s) { std::srand(s); }

}  // extern "C"
