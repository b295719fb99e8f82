// OUT-OF-TREE elementwise operations written the way a user of the reference writes them: a functor struct derived from
// prost::ElemOperation<DIM, COEFFS_COUNT[, SHARED_MEM_TYPE]> (elem_operation.hpp:30-40) with
//     operator()(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& tau_diag, T tau_scal, bool invert_tau),
// turned into a prox by prost::ProxElemOperation<T, OP> (prox_elem_operation.hpp:32-110) and registered by name as
// custom.cpp:11-17 does.  None of this is part of libprost.so: the generic gfx950 kernel is instantiated HERE, from
// include/prost/prox/prox_elem_operation.inl, by hipcc.
//
//   test:op:norm2_huber   user-written  c huber_alpha(a |x| - b) + d |x| + e/2 |x|^2   (7 coefficients, run-time dim);
//                         inside the library: elem_operation:norm2:huber
//   test:op:abs_1d        user-written  c |a x - b| + d x + e/2 x^2  (DIM = 1);   library: elem_operation:1d:abs
//   test:op:simplex_lds   projection onto the unit simplex, sorting in the per-thread LDS slice (SharedMem, dim entries of T,
//                         no coefficients);   library: elem_operation:ind_simplex
//   test:op:partial       writes res[0] only (kPartialResult): the other components keep their old content
//   test:op:partial_undeclared  the same operation WITHOUT the declaration (a straight port): the register-tile path preloads res
//   test:tpl:ind_sum / test:tpl:ind_simplex   the PUBLIC ElemOperationIndSum / ElemOperationIndSimplex (elemop/*.hpp), instantiated out of
//                         tree;   library: elem_operation:ind_sum / elem_operation:ind_simplex
//   test:tpl:1d:<fn> / test:tpl:norm2:<fn>   the PUBLIC templates ElemOperation1D / ElemOperationNorm2 over the 14 public
//                         Function1D* functors (elemop/*.hpp), instantiated out of tree;   library: elem_operation:1d|norm2:<fn>
#include <hip/hip_runtime.h>

#include "prost/factory.hpp"
#include "prost/prox/elemop/elem_operation.hpp"
#include "prost/prox/elemop/elem_operation_1d.hpp"
#include "prost/prox/elemop/elem_operation_ind_simplex.hpp"
#include "prost/prox/elemop/elem_operation_ind_sum.hpp"
#include "prost/prox/elemop/elem_operation_norm2.hpp"
#include "prost/prox/prox_elem_operation.inl"

namespace {

using prost::SharedMem;
using prost::Vector;

// ---- user-written norm2 + Huber, the formulas as a user would type them from the paper / the reference's sources ----
template <typename T>
struct Norm2Huber : public prost::ElemOperation<0, 7> {
  static const bool kWritesAllComponents = true;          // opt-in: every component is assigned below, the tile path skips the preload of res
  __host__ __device__ Norm2Huber(T* coeffs, size_t dim, SharedMem<SharedMemType, GetSharedMemCount>& shared_mem) : coeffs_(coeffs), dim_(dim) {}

  inline __host__ __device__ void operator()(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& tau_diag, T tau_scal, bool invert_tau) {
    T norm = 0;
    for (size_t i = 0; i < dim_; i++) {
      const T val = arg[i];
      norm += val * val;
    }
    if (norm > 0) {
      norm = sqrt(norm);
      T tau = invert_tau ? (1. / (tau_scal * tau_diag[0])) : (tau_scal * tau_diag[0]);
      const T prox_arg = ((coeffs_[0] * (norm - coeffs_[3] * tau)) / (1. + tau * coeffs_[4])) - coeffs_[1];
      const T step = (coeffs_[2] * coeffs_[0] * coeffs_[0] * tau) / (1. + tau * coeffs_[4]);
      // Huber with parameter alpha = coeffs_[5]
      T h = (prox_arg / step) / (1. + coeffs_[5] / step);
      h /= max(static_cast<T>(1), abs(h));
      const T prox_result = ((prox_arg - step * h) + coeffs_[1]) / coeffs_[0];
      for (size_t i = 0; i < dim_; i++) res[i] = prox_result * arg[i] / norm;
    } else {
      for (size_t i = 0; i < dim_; i++) res[i] = 0;
    }
  }

 private:
  T* coeffs_;
  size_t dim_;
};

// ---- user-written 1-D soft thresholding with the full coefficient set ----
template <typename T>
struct Abs1D : public prost::ElemOperation<1, 7> {
  __host__ __device__ Abs1D(T* coeffs, size_t dim, SharedMem<SharedMemType, GetSharedMemCount>& shared_mem) : coeffs_(coeffs) {}

  inline __host__ __device__ void operator()(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& tau_diag, T tau_scal, bool invert_tau) {
    T tau = invert_tau ? (1. / (tau_scal * tau_diag[0])) : (tau_scal * tau_diag[0]);
    if (coeffs_[0] == 0 || coeffs_[2] == 0) {
      res[0] = (arg[0] - tau * coeffs_[3]) / (1 + tau * coeffs_[4]);
    } else {
      const T prox_arg = ((coeffs_[0] * (arg[0] - coeffs_[3] * tau)) / (1. + tau * coeffs_[4])) - coeffs_[1];
      const T step = (coeffs_[2] * coeffs_[0] * coeffs_[0] * tau) / (1. + tau * coeffs_[4]);
      T shrunk = 0;
      if (prox_arg >= step) shrunk = prox_arg - step;
      else if (prox_arg <= -step) shrunk = prox_arg + step;
      res[0] = (shrunk + coeffs_[1]) / coeffs_[0];
    }
  }

 private:
  T* coeffs_;
};

// ---- simplex projection that sorts in the operation's per-thread LDS slice (the SharedMem hook) ----
template <typename T>
struct SimplexLds : public prost::ElemOperation<0, 0, T> {
  struct GetSharedMemCount {
    inline __host__ __device__ size_t operator()(size_t dim) { return dim; }
  };
  __device__ SimplexLds(size_t dim, SharedMem<T, GetSharedMemCount>& shared_mem) : dim_(dim), sh_(shared_mem) {}

  inline __device__ void operator()(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& tau_diag, T tau_scal, bool invert_tau) {
    for (size_t i = 0; i < dim_; i++) sh_[i] = arg[i];
    // descending insertion sort with decreasing gaps
    const int gaps[6] = {132, 57, 23, 10, 4, 1};
    for (int k = 0; k < 6; k++) {
      const int gap = gaps[k];
      for (int i = gap; i < (int)dim_; i++) {
        const T temp = sh_[i];
        int j = i;
        for (; (j >= gap) && (sh_[j - gap] <= temp); j -= gap) sh_[j] = sh_[j - gap];
        sh_[j] = temp;
      }
    }
    bool found = false;
    T tmpsum = 0, tmax = 0;
    for (int ii = 1; ii <= (int)dim_ - 1; ii++) {
      tmpsum += sh_[ii - 1];
      tmax = (tmpsum - 1.) / (T)ii;
      if (tmax >= sh_[ii]) { found = true; break; }
    }
    if (!found) tmax = (tmpsum + sh_[dim_ - 1] - 1.0) / (T)dim_;
    for (size_t i = 0; i < dim_; i++) res[i] = max(arg[i] - tmax, static_cast<T>(0));
  }

 private:
  size_t dim_;
  SharedMem<T, GetSharedMemCount>& sh_;
};

// ---- an operation that leaves components unwritten on purpose ----
template <typename T>
struct FirstComponentOnly : public prost::ElemOperation<0, 0> {
  static const bool kPartialResult = true;
  __host__ __device__ FirstComponentOnly(size_t dim, SharedMem<SharedMemType, GetSharedMemCount>& shared_mem) {}
  inline __host__ __device__ void operator()(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& tau_diag, T tau_scal, bool invert_tau) {
    res[0] = 2 * arg[0];
  }
};

// ---- the same omission WITHOUT the declaration: what a straight port from the reference looks like (its kernel preserves the old
// content of res; here the register-tile path must, too -- kWritesAllComponents defaults to false) ----
template <typename T>
struct FirstComponentOnlyUndeclared : public prost::ElemOperation<0, 0> {
  __host__ __device__ FirstComponentOnlyUndeclared(size_t dim, SharedMem<SharedMemType, GetSharedMemCount>& shared_mem) {}
  inline __host__ __device__ void operator()(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& tau_diag, T tau_scal, bool invert_tau) {
    res[0] = 2 * arg[0];
  }
};

// ---- factories: data = {count, dim, interleaved[, {coefficients}]} like sum_1d.m:79-80 / sum_norm2.m:85-86 ----
template <typename T, class OP>
prost::Prox<T>* CreateWithCoeffs(size_t idx, size_t size, bool diagsteps, const prost_value* data) {
  const size_t count = (size_t)prost::GetScalarFromCell(data, 0), dim = (size_t)prost::GetScalarFromCell(data, 1);
  const bool interleaved = prost::GetScalarFromCell(data, 2) > 0.;
  std::array<std::vector<T>, OP::kCoeffsCount> coeffs;
  prost::GetCoefficients<T, OP::kCoeffsCount>(coeffs, prost::GetCell(data, 3), OP::kDim == 1 ? size : count);
  return new prost::ProxElemOperation<T, OP>(idx, count, dim, interleaved, diagsteps, coeffs);
}
template <typename T, class OP>
prost::Prox<T>* CreateNoCoeffs(size_t idx, size_t, bool diagsteps, const prost_value* data) {
  return new prost::ProxElemOperation<T, OP>(idx, (size_t)prost::GetScalarFromCell(data, 0), (size_t)prost::GetScalarFromCell(data, 1),
                                             prost::GetScalarFromCell(data, 2) > 0., diagsteps);
}

template <typename T, template <typename> class FUN>
void RegisterTemplates(const char* fn) {
  auto& reg = prost::Factory<T>::prox_reg();
  reg[std::string("test:tpl:1d:") + fn] = CreateWithCoeffs<T, prost::ElemOperation1D<T, FUN<T>>>;
  reg[std::string("test:tpl:norm2:") + fn] = CreateWithCoeffs<T, prost::ElemOperationNorm2<T, FUN<T>>>;
}
template <typename T>
void RegisterAll() {
  auto& reg = prost::Factory<T>::prox_reg();
  reg["test:op:norm2_huber"] = CreateWithCoeffs<T, Norm2Huber<T>>;
  reg["test:op:abs_1d"] = CreateWithCoeffs<T, Abs1D<T>>;
  reg["test:op:simplex_lds"] = CreateNoCoeffs<T, SimplexLds<T>>;
  reg["test:op:partial"] = CreateNoCoeffs<T, FirstComponentOnly<T>>;
  reg["test:op:partial_undeclared"] = CreateNoCoeffs<T, FirstComponentOnlyUndeclared<T>>;
  reg["test:tpl:ind_sum"] = CreateNoCoeffs<T, prost::ElemOperationIndSum<T>>;
  reg["test:tpl:ind_simplex"] = CreateNoCoeffs<T, prost::ElemOperationIndSimplex<T>>;
  RegisterTemplates<T, prost::Function1DZero>("zero");
  RegisterTemplates<T, prost::Function1DAbs>("abs");
  RegisterTemplates<T, prost::Function1DSquare>("square");
  RegisterTemplates<T, prost::Function1DIndLeq0>("ind_leq0");
  RegisterTemplates<T, prost::Function1DIndGeq0>("ind_geq0");
  RegisterTemplates<T, prost::Function1DIndEq0>("ind_eq0");
  RegisterTemplates<T, prost::Function1DIndBox01>("ind_box01");
  RegisterTemplates<T, prost::Function1DMaxPos0>("max_pos0");
  RegisterTemplates<T, prost::Function1DL0>("l0");
  RegisterTemplates<T, prost::Function1DHuber>("huber");
  RegisterTemplates<T, prost::Function1DLq>("lq");
  RegisterTemplates<T, prost::Function1DLqPlusEps>("lq_plus_eps");
  RegisterTemplates<T, prost::Function1DTruncLinear>("trunclin");
  RegisterTemplates<T, prost::Function1DTruncQuad>("truncquad");
}

const bool registered = [] {
  RegisterAll<float>();
  RegisterAll<double>();
  return true;
}();

}  // namespace
