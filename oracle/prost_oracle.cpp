// prost_oracle.cpp -- CPU ORACLE: a from-scratch restatement of the reference
// (tum-vision/prost) hot path.  TEST INFRASTRUCTURE ONLY (see prost_oracle.h).
//
// Every routine cites the reference file:line it restates.  Arithmetic follows the
// reference expression by expression, including the places where double literals
// promote fp32 sub-expressions to double (elem_operation_1d.hpp:40-51,
// elem_operation_norm2.hpp:61-68, function_1d.hpp:66,153, backend_pdhg.cu:485).
// Build with -ffp-contract=off so no FMA contraction changes roundings.
//
// Parity status: pinned against oracle/_ref (reference sources compiled where they
// lie) -- see tests/test_oracle_pinning.py; ADMM/CGLS unpinned (needs cuBLAS/cuSPARSE).

#include "prost_oracle.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <list>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif
#ifdef __linux__
#include <sched.h>
#include <fstream>
#endif

namespace {

thread_local std::string g_err;
int g_threads = 1;

struct OrcError : std::runtime_error { using std::runtime_error::runtime_error; };

#define ORC_TRY try {
#define ORC_CATCH } catch (const std::exception& e) { g_err = e.what(); return 1; } return 0;

#define PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(g_threads) if(g_threads > 1)")

typedef long long ssz;

// ---------------------------------------------------------------------------------
// Multi-threaded TIMING runs only (bench.py's cpu_baseline; results do not depend on it):
// where the pages of the big vectors live and which cores the OpenMP threads run on.
// A 2-socket host streams 10x faster from node-local memory; std::vector::assign touches
// every page from the calling thread, i.e. puts the whole state on one node.
// ---------------------------------------------------------------------------------
// re-home a filled vector: the thread that will stream range [t n / T, (t+1) n / T) under PAR_FOR's static schedule
// touches it first
template <class V> void numa_rehome(V& v) {
  typedef typename V::value_type E;
  if (g_threads <= 1 || v.size() < ((size_t)1 << 18)) return;
  V w; w.reserve(v.size());
  E* p = w.data(); const E* src = v.data(); const ssz n = (ssz)v.size();
  PAR_FOR
  for (ssz i = 0; i < n; i++) p[i] = src[i];          // first touch (raw reserved storage; E is trivially copyable)
  w.assign(v.begin(), v.end());                        // sets the size; the pages stay where they are
  v.swap(w);
}

#ifdef __linux__
// logical CPUs of this process's affinity mask: one hardware thread per physical core first (cores in (package, core id) order),
// then their SMT siblings in the same order; *cores = number of physical cores
std::vector<int> cpu_order(int* cores) {
  cpu_set_t mask; CPU_ZERO(&mask);
  *cores = 0;
  if (sched_getaffinity(0, sizeof(mask), &mask) != 0) return {};
  struct Ent { int pkg, core, cpu; };
  std::vector<Ent> all;
  for (int c = 0; c < CPU_SETSIZE; c++) {
    if (!CPU_ISSET(c, &mask)) continue;
    int pkg = 0, core = c;
    { std::ifstream f("/sys/devices/system/cpu/cpu" + std::to_string(c) + "/topology/physical_package_id"); if (f) f >> pkg; }
    { std::ifstream f("/sys/devices/system/cpu/cpu" + std::to_string(c) + "/topology/core_id"); if (f) f >> core; }
    all.push_back({pkg, core, c});
  }
  std::sort(all.begin(), all.end(), [](const Ent& a, const Ent& b) { return std::tie(a.pkg, a.core, a.cpu) < std::tie(b.pkg, b.core, b.cpu); });
  std::vector<int> first, rest;
  for (size_t i = 0; i < all.size(); i++) {
    const bool sibling = i > 0 && all[i].pkg == all[i - 1].pkg && all[i].core == all[i - 1].core;
    (sibling ? rest : first).push_back(all[i].cpu);
  }
  *cores = (int)first.size();
  first.insert(first.end(), rest.begin(), rest.end());
  return first;
}
cpu_set_t g_saved_mask; bool g_mask_saved = false;
// on = 1: pin OpenMP thread t of a g_threads-wide team to ONE CPU -- T <= physical cores: spread evenly over the cores in topology
// order (neighbouring static-schedule ranges on neighbouring cores of one node); more: fill the SMT siblings too; more threads than
// CPUs: wrap.  on = 0: every thread of the team (and the caller) gets the mask back that the process had.  -> threads bound
int bind_threads(int on) {
  if (!g_mask_saved) { CPU_ZERO(&g_saved_mask); if (sched_getaffinity(0, sizeof(g_saved_mask), &g_saved_mask) != 0) return 0; g_mask_saved = true; }
  const cpu_set_t saved = g_saved_mask;
  sched_setaffinity(0, sizeof(saved), &saved);
  int cores = 0;
  const std::vector<int> order = on ? cpu_order(&cores) : std::vector<int>();
  const int T = g_threads, C = (int)order.size();
  int bound = 0;
#pragma omp parallel num_threads(g_threads) reduction(+ : bound)
  {
    if (!on || C == 0) sched_setaffinity(0, sizeof(saved), &saved);
    else {
#ifdef _OPENMP
      const int t = omp_get_thread_num();
#else
      const int t = 0;
#endif
      const int span = T <= cores ? cores : C;                       // the CPUs this team spreads over
      const int slot = (int)(((long long)t * span) / T) % C;
      cpu_set_t one; CPU_ZERO(&one); CPU_SET(order[slot], &one);
      if (sched_setaffinity(0, sizeof(one), &one) == 0) bound += 1;
    }
  }
  return bound;
}
#else
int bind_threads(int) { return 0; }
#endif

// ---------------------------------------------------------------------------------
// Sums whose ORDER the reference leaves unspecified -- thrust::transform_reduce on the device (cgls.hpp:152-170 NormSquared,
// problem.cu:456-486 inside normest) and cublas<t>nrm2 (backend_admm.cu:40-50) reduce by trees nobody documents; the host build
// of thrust happens to go left to right.  The CG scalars alpha / beta, the relative stopping test of cgls.hpp:326-360 and the
// rescale of the initial step sizes by normest are knife-edge functions of those sums, so "some order" is not a contract two
// implementations can share.  The oracle therefore takes the one value every order approximates: the EXACT sum of the terms,
// rounded once -- accumulated as an unevaluated pair hi + lo (Knuth's TwoSum keeps the rounding error of every addition; the
// pair is exact to ~2^-100 of the sum).  prost_amd/csrc/reduce.hpp accumulates the same way in every kernel variant, which is
// what makes ADMM / CGLS and normest comparable bit for bit.  The value stays within the rounding error of any summation
// order; the pins of DESIGN.md section 2 (normest against the compiled reference: rel 1e-6) are unchanged.
// Needs -ffp-contract=off (oracle/Makefile).
// ---------------------------------------------------------------------------------
struct ExactSum {
  double hi = 0., lo = 0.;
  inline void add(double t) {
    const double s = hi + t;
    const double bb = s - hi;
    const double e = (hi - (s - bb)) + (t - bb);
    hi = s; lo += e;
  }
  inline double value() const { return hi + lo; }
};
template <class T> inline double exact_sum_sq(const T* v, size_t n) {
  ExactSum a;
  for (size_t i = 0; i < n; i++) a.add((double)v[i] * (double)v[i]);
  return a.value();
}

// ---------------------------------------------------------------------------------
// glibc rand() (TYPE_3, r[i] = r[i-3] + r[i-31]); normest seeds from std::rand()
// with srand never called (problem.cu:435) -> seed 1 in a fresh process.
// ---------------------------------------------------------------------------------
struct GlibcRand {
  std::vector<uint32_t> r; size_t k;
  explicit GlibcRand(unsigned seed) : r(344), k(0) {
    if (seed == 0) seed = 1;
    r[0] = seed;
    for (int i = 1; i < 31; i++) {
      long long hi = (int32_t)r[i - 1] / 127773, lo = (int32_t)r[i - 1] % 127773;
      long long w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      r[i] = (uint32_t)w;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    for (int i = 34; i < 344; i++) r[i] = r[i - 31] + r[i - 3];
  }
  int32_t next() {
    uint32_t v = r[r.size() - 31] + r[r.size() - 3];
    r.push_back(v);
    if (r.size() > 4096) r.erase(r.begin(), r.begin() + 2048);
    return (int32_t)(v >> 1);
  }
};

// ---------------------------------------------------------------------------------
// Function1D*  (include/prost/prox/elemop/function_1d.hpp:34-326)
// ---------------------------------------------------------------------------------
template <class T> inline T fn_abs(T x0, T tau) {            // :47-60
  if (x0 >= tau) return x0 - tau;
  else if (x0 <= -tau) return x0 + tau;
  return 0;
}
template <class T> inline T fn_square(T x0, T tau) {         // :63-72
  return (T)((double)x0 / (1. + (double)tau));
}
template <class T> inline T fn_l0(T x0, T tau) {             // :146-158
  if (x0 * x0 > 2 * tau) return x0;
  return 0;
}
template <class T> inline T lq_newton(const T t0, const T alpha, const T q, const T eps) {  // :173-191
  T t = t0, delta = 0;
  do {
    const T power = std::pow(t, q);
    const T dF1 = t - 1 + alpha * q * power / t;
    const T dF2 = 1 + alpha * q * (q - 1) * power / (t * t);
    delta = dF1 / dF2;
    t = t - delta;
  } while (delta > eps);
  return t;
}
template <class T> inline T lq_half(const T alpha) {         // :195-202
  const T sqrt3 = std::sqrt(static_cast<T>(3));
  const T PI_half = static_cast<T>(1.5707963267948966192313216916397514420985846996875529);
  const T s = 2 * (std::sin(static_cast<T>((std::acos(static_cast<T>(alpha * 3 * sqrt3 / 4)) + PI_half) / 3))) / sqrt3;
  return s * s;
}
template <class T> inline T lq_eps();
template <> inline float lq_eps<float>() { return (float)1e-5; }     // :263-267
template <> inline double lq_eps<double>() { return 1e-11; }          // :270-274

template <class T> __attribute__((always_inline)) inline T fn_apply(int fn, T x0, T tau, T alpha, T beta) {
  switch (fn) {
    case ORC_FN_ZERO: return x0;                                        // :34-44
    case ORC_FN_ABS: return fn_abs(x0, tau);
    case ORC_FN_SQUARE: return fn_square(x0, tau);
    case ORC_FN_IND_LEQ0: if ((double)x0 > 0.) return (T)0.; return x0;  // :75-87
    case ORC_FN_IND_GEQ0: if ((double)x0 < 0.) return (T)0.; return x0;  // :90-102
    case ORC_FN_IND_EQ0: return (T)0.;                                   // :105-114
    case ORC_FN_IND_BOX01:                                               // :117-131
      if ((double)x0 > 1.) return (T)1.;
      else if ((double)x0 < 0.) return (T)0.;
      return x0;
    case ORC_FN_MAX_POS0:                                                // :134-148
      if (x0 > tau) return x0 - tau;
      else if ((double)x0 < 0.) return x0;
      return (T)0.;
    case ORC_FN_L0: return fn_l0(x0, tau);
    case ORC_FN_HUBER: {                                                 // :161-171
      T result = (T)(((double)(x0 / tau)) / (1. + (double)(alpha / tau)));
      result /= std::max(static_cast<T>(1), std::abs(result));
      return x0 - tau * result;
    }
    case ORC_FN_LQ: {                                                    // :205-260
      if (alpha == 1) return fn_abs(x0, tau);
      else if (alpha == 0) return fn_l0(x0, tau);
      T t = 0;
      if (std::abs(x0) > 0) {
        T factor = tau * std::pow(std::abs(x0), static_cast<T>(alpha - 2));
        if (alpha < 1) {
          const T t2 = 2 * (alpha - 1) / (alpha - 2);
          if ((double)factor < 0.5 * (double)(1 - (t2 - 1) * (t2 - 1)) / (double)std::pow(t2, alpha)) {
            if ((double)alpha == 0.5) t = lq_half<T>(factor);
            else t = lq_newton<T>(1, factor, alpha, lq_eps<T>());
          }
        } else {
          t = lq_newton<T>(1, factor, alpha, lq_eps<T>());
        }
      }
      return t * std::abs(x0);
    }
    case ORC_FN_LQ_PLUS_EPS: return 0;                                   // :294-306 (stub)
    case ORC_FN_TRUNCQUAD: {                                             // :277-291
      const T x_sq = fn_square<T>(x0, 2 * tau * alpha);
      const T en_sq = alpha * x_sq * x_sq + (x_sq - x0) * (x_sq - x0) / (2 * tau);
      if (en_sq < beta) return x_sq;
      return x0;
    }
    case ORC_FN_TRUNCLIN: {                                              // :309-323
      const T x_shrink = fn_abs<T>(x0, tau * alpha);
      const T en_shrink = (x_shrink - x0) * (x_shrink - x0) / (2 * tau) + alpha * std::abs(x_shrink);
      if (en_shrink < beta) return x_shrink;
      return x0;
    }
  }
  throw OrcError("unknown function id");
}

// Vector view: include/prost/prox/vector.hpp:44-48
struct View {
  size_t count, dim; bool interleaved; size_t tx;
  inline size_t at(size_t i) const { return interleaved ? (tx * dim + i) : (tx + count * i); }
};

// ElemOperation1D::operator()  (elem_operation_1d.hpp:36-59)
template <class T>
__attribute__((always_inline)) inline void elem_1d(int fn, T* res, const T* arg, const T* tau_diag, T tau_scal, bool invert_tau,
                    const T* c, const View& v) {
  const size_t i0 = v.at(0);
  T tau = invert_tau ? (T)(1. / (double)(tau_scal * tau_diag[i0])) : (tau_scal * tau_diag[i0]);
  if (c[0] == 0 || c[2] == 0) {
    res[i0] = (arg[i0] - tau * c[3]) / (1 + tau * c[4]);
  } else {
    const T prox_arg = (T)(((double)(c[0] * (arg[i0] - c[3] * tau)) / (1. + (double)(tau * c[4]))) - (double)c[1]);
    const T step = (T)((double)(c[2] * c[0] * c[0] * tau) / (1. + (double)(tau * c[4])));
    res[i0] = (fn_apply<T>(fn, prox_arg, step, c[5], c[6]) + c[1]) / c[0];
  }
}

// ElemOperationNorm2::operator()  (elem_operation_norm2.hpp:40-88)
template <class T>
__attribute__((always_inline)) inline void elem_norm2(int fn, T* res, const T* arg, const T* tau_diag, T tau_scal, bool invert_tau,
                       const T* c, const View& v) {
  T norm = 0;
  for (size_t i = 0; i < v.dim; i++) { const T val = arg[v.at(i)]; norm += val * val; }
  if (norm > 0) {
    norm = std::sqrt(norm);
    const size_t i0 = v.at(0);
    T tau = invert_tau ? (T)(1. / (double)(tau_scal * tau_diag[i0])) : (tau_scal * tau_diag[i0]);
    const T prox_arg = (T)(((double)(c[0] * (norm - c[3] * tau)) / (1. + (double)(tau * c[4]))) - (double)c[1]);
    const T step = (T)((double)(c[2] * c[0] * c[0] * tau) / (1. + (double)(tau * c[4])));
    const T prox_result = (fn_apply<T>(fn, prox_arg, step, c[5], c[6]) + c[1]) / c[0];
    for (size_t i = 0; i < v.dim; i++) res[v.at(i)] = prox_result * arg[v.at(i)] / norm;
  } else {
    for (size_t i = 0; i < v.dim; i++) res[v.at(i)] = 0;
  }
}

// ProxElemOperationKernel (prox_elem_operation.inl:59-94): one "thread" per tx.  The reference instantiates the kernel per
// (operation, function) pair (prox_elem_operation.inl:96-198); so does this loop -- FN, OP, the layout and small dims are
// compile-time constants, which lets the compiler fold the function switch and unroll the component loops (same expressions,
// same results; only the speed of the CPU baseline depends on it).
template <class T, int FN, int OP, int DIM, bool IL>
void prox_elem_loop(T* res, const T* arg, const T* tau_diag, T tau, bool invert, size_t count, size_t dim_rt,
                    const T* const* cptr, const T* cval) {
  const size_t dim = DIM > 0 ? (size_t)DIM : dim_rt;
  PAR_FOR
  for (ssz t = 0; t < (ssz)count; t++) {
    View v{count, dim, IL, (size_t)t};
    T c[7];
    for (int i = 0; i < 7; i++) c[i] = cptr[i] ? cptr[i][t] : cval[i];
    if (OP == ORC_OP_1D) elem_1d<T>(FN, res, arg, tau_diag, tau, invert, c, v);
    else elem_norm2<T>(FN, res, arg, tau_diag, tau, invert, c, v);
  }
}
template <class T, int FN>
void prox_elem_fn(int op, T* res, const T* arg, const T* tau_diag, T tau, bool invert, size_t count, size_t dim, bool interleaved,
                  const T* const* cptr, const T* cval) {
  if (op == ORC_OP_1D) { prox_elem_loop<T, FN, ORC_OP_1D, 1, false>(res, arg, tau_diag, tau, invert, count, 1, cptr, cval); return; }
#define ORC_N2(D) (interleaved ? prox_elem_loop<T, FN, ORC_OP_NORM2, D, true>(res, arg, tau_diag, tau, invert, count, dim, cptr, cval) \
                               : prox_elem_loop<T, FN, ORC_OP_NORM2, D, false>(res, arg, tau_diag, tau, invert, count, dim, cptr, cval))
  switch (dim) { case 2: ORC_N2(2); break; case 3: ORC_N2(3); break; case 4: ORC_N2(4); break; default: ORC_N2(0); }
#undef ORC_N2
}
template <class T, int... FN>
void prox_elem_dispatch(std::integer_sequence<int, FN...>, int op, int fn, T* res, const T* arg, const T* tau_diag, T tau, bool invert,
                        size_t count, size_t dim, bool interleaved, const T* const* cptr, const T* cval) {
  typedef void (*Fn)(int, T*, const T*, const T*, T, bool, size_t, size_t, bool, const T* const*, const T*);
  static const Fn table[] = {&prox_elem_fn<T, FN>...};
  table[fn](op, res, arg, tau_diag, tau, invert, count, dim, interleaved, cptr, cval);
}
template <class T>
void prox_elem_run(int op, int fn, T* res, const T* arg, const T* tau_diag, T tau, bool invert,
                   size_t count, size_t dim, bool interleaved, const T* const* cptr, const T* cval) {
  if (fn < 0 || fn >= ORC_FN_COUNT) throw OrcError("unknown function id");
  if (op == ORC_OP_1D) dim = 1;   // kDim = 1 for ElemOperation1D (elem_operation_1d.hpp:30)
  prox_elem_dispatch<T>(std::make_integer_sequence<int, ORC_FN_COUNT>(), op, fn, res, arg, tau_diag, tau, invert, count, dim, interleaved, cptr, cval);
}

// helper::ProjectEpiQuadNd (include/prost/prox/helper.hpp:44-105); x0 may alias x
template <class T>
inline void project_epi_quad_nd(T* d, const View& vx, const T y0, const T alpha, T& y, size_t dim) {
  T sq_norm_x0 = static_cast<T>(0);
  for (size_t i = 0; i < dim; i++) sq_norm_x0 += d[vx.at(i)] * d[vx.at(i)];
  const T norm_x0 = std::sqrt(sq_norm_x0);
  if (y0 >= alpha * sq_norm_x0) { y = y0; return; }
  const T a = (T)(2. * (double)alpha * (double)norm_x0);
  const T b = (T)(2. * (1. - 2. * (double)alpha * (double)y0) / 3.);
  T dd, v;
  if (b < 0) {
    const T sq = std::pow(-b, static_cast<T>(3. / 2.));
    dd = (a - sq) * (a + sq);
  } else {
    dd = a * a + b * b * b;
  }
  if (dd >= 0) {
    const T c = std::pow(a + std::sqrt(dd), static_cast<T>(1. / 3.));
    if ((double)std::abs(c) > 1e-6) v = c - b / c;
    else v = 0;
  } else {
    v = 2 * std::sqrt(-b) * std::cos(std::acos(a / std::pow(-b, static_cast<T>(3. / 2.))) / static_cast<T>(3.));
  }
  if (norm_x0 > 0) {
    for (size_t i = 0; i < dim; i++)
      d[vx.at(i)] = (T)(((double)v / (2. * (double)alpha)) * (double)(d[vx.at(i)] / norm_x0));
  } else {
    for (size_t i = 0; i < dim; i++) d[vx.at(i)] = 0;
  }
  T sq_norm_x = static_cast<T>(0);
  for (size_t i = 0; i < dim; i++) sq_norm_x += d[vx.at(i)] * d[vx.at(i)];
  y = alpha * sq_norm_x;
}

// ProxIndEpiQuadKernel (src/prox/prox_ind_epi_quad.cu:42-79)
template <class T>
void epi_quad_run(T* res, const T* arg, size_t count, size_t dim,
                  const T* a_ptr, T a_val, const T* b_ptr, const T* c_ptr, T c_val) {
  PAR_FOR
  for (ssz t = 0; t < (ssz)count; t++) {
    View vx{count, dim - 1, false, (size_t)t};
    T& y = res[count * (dim - 1) + t];
    const T y0 = arg[count * (dim - 1) + t];
    const T a = a_ptr ? a_ptr[t] : a_val;
    const T c = c_ptr ? c_ptr[t] : c_val;
    T sq_norm_b = static_cast<T>(0);
    for (size_t i = 0; i < dim - 1; i++) {
      T val = b_ptr[vx.at(i)];
      res[vx.at(i)] = arg[vx.at(i)] + (val / (2 * a));
      sq_norm_b += val * val;
    }
    project_epi_quad_nd<T>(res, vx, y0 - c + (sq_norm_b / (4 * a)), a, y, dim - 1);
    for (size_t i = 0; i < dim - 1; i++) res[vx.at(i)] -= b_ptr[vx.at(i)] / (2 * a);
    y = y + c - (sq_norm_b / (4 * a));
  }
}

// ---------------------------------------------------------------------------------
// Linear operator blocks
// ---------------------------------------------------------------------------------
// BlockGradient2DKernel / Adjoint (src/linop/block_gradient2d.cu:26-78, :81-139).  The kernel decodes (x, y, l) from the thread
// index; here the same elements are visited by nested loops (no division per element), the contiguous index innermost.
template <class T>
void grad2d_run(bool adjoint, T* res, const T* rhs, size_t nx, size_t ny, size_t L, bool lf) {
  const size_t N = nx * ny * L;
  const size_t sy = lf ? L : 1, sx = lf ? ny * L : ny, sl = lf ? 1 : nx * ny;
  const size_t n_out = lf ? ny : L, n_in = lf ? L : ny;          // label_first: l fastest; else y fastest
  PAR_FOR
  for (ssz xs = 0; xs < (ssz)nx; xs++) {
    const size_t x = (size_t)xs;
    for (size_t o = 0; o < n_out; o++) {
      for (size_t in = 0; in < n_in; in++) {
        const size_t y = lf ? o : in, l = lf ? in : o;
        const size_t idx = y * sy + x * sx + l * sl;
        if (!adjoint) {
          const T val_pt = rhs[idx];
          T gx, gy;
          if (y < ny - 1) gy = rhs[idx + sy] - val_pt; else gy = 0;
          if (x < nx - 1) gx = rhs[idx + sx] - val_pt; else gx = 0;
          res[idx] += gx;
          res[idx + N] += gy;
        } else {
          T divx, divy;
          if (y < ny - 1) divy = rhs[idx + N]; else divy = 0;
          if (y > 0) divy -= rhs[idx + N - sy];
          if (x < nx - 1) divx = rhs[idx]; else divx = 0;
          if (x > 0) divx -= rhs[idx - sx];
          res[idx] -= (divx + divy);
        }
      }
    }
  }
}

// BlockGradient3DKernel / Adjoint (src/linop/block_gradient3d.cu:25-81, :83-150)
template <class T>
void grad3d_run(bool adjoint, T* res, const T* rhs, size_t nx, size_t ny, size_t L, bool lf) {
  const size_t N = nx * ny * L;
  const size_t sy = lf ? L : 1, sx = lf ? ny * L : ny, sl = lf ? 1 : nx * ny;
  const size_t n_out = lf ? ny : L, n_in = lf ? L : ny;
  PAR_FOR
  for (ssz xs = 0; xs < (ssz)nx; xs++) {
    const size_t x = (size_t)xs;
    for (size_t o = 0; o < n_out; o++) {
      for (size_t in = 0; in < n_in; in++) {
        const size_t y = lf ? o : in, l = lf ? in : o;
        const size_t idx = y * sy + x * sx + l * sl;
        if (!adjoint) {
          T gx = 0, gy = 0, gl = 0;
          const T val_pt = rhs[idx];
          if (y < ny - 1) gy = rhs[idx + sy] - val_pt;
          if (x < nx - 1) gx = rhs[idx + sx] - val_pt;
          if (l < L - 1) gl = rhs[idx + sl] - val_pt; else gl = -val_pt;   // dirichlet :73-76
          res[idx] += gx;
          res[idx + N] += gy;
          res[idx + 2 * N] += gl;
        } else {
          T divx = 0, divy = 0, divl = 0;
          if (y < ny - 1) divy = rhs[idx + N]; else divy = 0;
          if (y > 0) divy -= rhs[idx + N - sy];
          if (x < nx - 1) divx = rhs[idx]; else divx = 0;
          if (x > 0) divx -= rhs[idx - sx];
          divl = rhs[idx + 2 * N];
          if (l > 0) divl -= rhs[idx + 2 * N - sl];
          res[idx] -= (divx + divy + divl);
        }
      }
    }
  }
}

// BlockDiagsKernel / AdjointKernel (src/linop/block_diags.cu:36-96)
template <class T>
void diags_run(bool adjoint, T* res, const T* rhs, size_t nrows, size_t ncols, size_t ndiags,
               const int64_t* ofs, const float* fac, bool quirk) {
  if (!adjoint) {
    PAR_FOR
    for (ssz r = 0; r < (ssz)nrows; r++) {
      T result = 0;
      for (size_t i = 0; i < ndiags; i++) {
        const ssz col = r + ofs[i];
        if (col < 0) continue;
        if (col >= (ssz)ncols) break;
        result += rhs[col] * fac[i];
      }
      res[r] += result;
    }
  } else {
    // launch grid from nrows (block_diags.cu:210-211): threads = ceil(nrows/256)*256
    size_t limit = ncols;
    if (quirk) limit = std::min(ncols, ((nrows + 255) / 256) * 256);
    PAR_FOR
    for (ssz col = 0; col < (ssz)limit; col++) {
      T result = 0;
      for (size_t i = 0; i < ndiags; i++) {
        ssz o = ofs[i];
        if (o <= col && (col - o) < (ssz)nrows && (col - o) >= 0) result += rhs[col - o] * fac[i];
        if (o > col) break;
      }
      res[col] += result;
    }
  }
}

// csr2csc (src/common.cu:55-82)
template <class T>
void csr2csc_run(int n, int m, int nz, const T* a, const int32_t* col_idx, const int32_t* row_start,
                 T* csc_a, int32_t* row_idx, int32_t* col_start) {
  for (int i = 0; i <= m; i++) col_start[i] = 0;
  for (int i = 0; i < nz; i++) col_start[col_idx[i] + 1]++;
  for (int i = 0; i < m; i++) col_start[i + 1] += col_start[i];
  for (int i = 0; i < n; i++)
    for (int j = row_start[i]; j < row_start[i + 1]; j++) {
      int k = col_idx[j];
      int l = col_start[k]++;
      row_idx[l] = i;
      if (a) csc_a[l] = a[j];
    }
  for (int i = m; i > 0; i--) col_start[i] = col_start[i - 1];
  col_start[0] = 0;
}

// y = 1*A*x + 1*y, CSR (cusparse<t>csrmv as called at block_sparse.cu:156-168).
// cuSPARSE 10.2 is not under /root/reference; textbook row-wise dot product, row sum
// accumulated in T in index order.
template <class T>
void csr_spmv_acc(T* res, const T* rhs, int nrows, const T* val, const int32_t* ptr, const int32_t* ind) {
  PAR_FOR
  for (int r = 0; r < nrows; r++) {
    T sum = 0;
    for (int32_t j = ptr[r]; j < ptr[r + 1]; j++) sum += val[j] * rhs[ind[j]];
    res[r] += sum;
  }
}

// BlockSparseKronIdKernel (block_sparse_kron_id.cu:26-49) / BlockIdKronSparseKernel (block_id_kron_sparse.cu:26-52):
// one output element per thread, matrix values held as float whatever T is
template <class T>
void kron_spmv_acc(bool id_first, T* res, const T* rhs, size_t diaglength, size_t nrows, size_t ncols, const float* val,
                   const int32_t* ptr, const int32_t* ind) {
  PAR_FOR
  for (ssz tx = 0; tx < (ssz)(diaglength * nrows); tx++) {
    size_t row, col_ofs;
    if (id_first) { row = (size_t)tx % nrows; col_ofs = ((size_t)tx / nrows) * ncols; }
    else { col_ofs = (size_t)tx % diaglength; row = (size_t)tx / diaglength; }
    T sum = 0;
    for (int32_t i = ptr[row]; i < ptr[row + 1]; i++)
      sum += val[i] * rhs[id_first ? (size_t)ind[i] + col_ofs : (size_t)ind[i] * diaglength + col_ofs];
    res[tx] += sum;
  }
}

enum { BK_GRAD2D, BK_GRAD3D, BK_DIAGS, BK_SPARSE, BK_ZERO, BK_SPARSE_KRON_ID, BK_ID_KRON_SPARSE };

template <class T>
struct Block {
  int kind; size_t row, col, nrows, ncols;
  size_t nx = 0, ny = 0, L = 0; bool lf = false;
  size_t ndiags = 0; std::vector<int64_t> ofs; std::vector<float> fac;
  int nnz = 0;
  std::vector<T> val, val_t; std::vector<int32_t> ptr, ind, ptr_t, ind_t;   // K (CSR), K^T (CSR)
  std::vector<float> fval, fval_t; size_t diaglength = 0; int mat_nrows = 0, mat_ncols = 0;   // Kronecker blocks
  void rehome() { numa_rehome(val); numa_rehome(val_t); numa_rehome(ptr); numa_rehome(ind); numa_rehome(ptr_t); numa_rehome(ind_t); numa_rehome(fval); numa_rehome(fval_t); }

  void add(T* res, const T* rhs) const {              // EvalLocalAdd
    switch (kind) {
      case BK_GRAD2D: grad2d_run<T>(false, res, rhs, nx, ny, L, lf); break;
      case BK_GRAD3D: grad3d_run<T>(false, res, rhs, nx, ny, L, lf); break;
      case BK_DIAGS: diags_run<T>(false, res, rhs, nrows, ncols, ndiags, ofs.data(), fac.data(), false); break;
      case BK_SPARSE: csr_spmv_acc<T>(res, rhs, (int)nrows, val.data(), ptr.data(), ind.data()); break;
      case BK_SPARSE_KRON_ID: case BK_ID_KRON_SPARSE:
        kron_spmv_acc<T>(kind == BK_ID_KRON_SPARSE, res, rhs, diaglength, mat_nrows, mat_ncols, fval.data(), ptr.data(), ind.data()); break;
      case BK_ZERO: break;
    }
  }
  void add_adj(T* res, const T* rhs, bool quirk) const {   // EvalAdjointLocalAdd
    switch (kind) {
      case BK_GRAD2D: grad2d_run<T>(true, res, rhs, nx, ny, L, lf); break;
      case BK_GRAD3D: grad3d_run<T>(true, res, rhs, nx, ny, L, lf); break;
      case BK_DIAGS: diags_run<T>(true, res, rhs, nrows, ncols, ndiags, ofs.data(), fac.data(), quirk); break;
      case BK_SPARSE: csr_spmv_acc<T>(res, rhs, (int)ncols, val_t.data(), ptr_t.data(), ind_t.data()); break;
      case BK_SPARSE_KRON_ID: case BK_ID_KRON_SPARSE:
        kron_spmv_acc<T>(kind == BK_ID_KRON_SPARSE, res, rhs, diaglength, mat_ncols, mat_nrows, fval_t.data(), ptr_t.data(), ind_t.data()); break;
      case BK_ZERO: break;
    }
  }
  T row_sum(size_t r, T alpha) const {
    switch (kind) {
      case BK_GRAD2D: case BK_GRAD3D: return 2;                       // block_gradient2d.cu:154-157, 3d :165-168
      case BK_DIAGS: {                                                 // block_diags.cu:121-136
        T sum = 0;
        for (size_t i = 0; i < ndiags; i++) {
          const ssz c = (ssz)r + ofs[i];
          if (c < 0) continue;
          if ((size_t)c >= ncols) break;
          sum += std::pow(std::abs(fac[i]), alpha);
        }
        return sum;
      }
      case BK_SPARSE: {                                                // block_sparse.cu:112-120
        T sum = 0;
        for (int32_t i = ptr[r]; i < ptr[r + 1]; i++) sum += std::pow(std::abs(val[i]), alpha);
        return sum;
      }
      case BK_SPARSE_KRON_ID: case BK_ID_KRON_SPARSE: {                // block_sparse_kron_id.cu:118-128, block_id_kron_sparse.cu:127-137
        r = kind == BK_SPARSE_KRON_ID ? r / diaglength : r % (size_t)mat_nrows;
        T sum = 0;
        for (int32_t i = ptr[r]; i < ptr[r + 1]; i++) sum += std::pow(std::abs(fval[i]), alpha);
        return sum;
      }
    }
    return 0;                                                          // block_zero.cu
  }
  T col_sum(size_t c, T alpha) const {
    switch (kind) {
      case BK_GRAD2D: return 4;                                        // block_gradient2d.cu:160-163
      case BK_GRAD3D: return 6;                                        // block_gradient3d.cu:171-174
      case BK_DIAGS: {                                                 // block_diags.cu:138-162
        T sum = 0; ssz sc = (ssz)c;
        for (size_t i = 0; i < ndiags; i++) {
          ssz o = ofs[i];
          if (o <= sc && (sc - o) < (ssz)nrows && (sc - o) >= 0) sum += std::pow(std::abs(fac[i]), alpha);
          if (o > sc) break;
        }
        return sum;
      }
      case BK_SPARSE: {                                                // block_sparse.cu:123-131
        T sum = 0;
        for (int32_t i = ptr_t[c]; i < ptr_t[c + 1]; i++) sum += std::pow(std::abs(val_t[i]), alpha);
        return sum;
      }
      case BK_SPARSE_KRON_ID: case BK_ID_KRON_SPARSE: {                // block_sparse_kron_id.cu:130-140, block_id_kron_sparse.cu:139-149
        c = kind == BK_SPARSE_KRON_ID ? c / diaglength : c % (size_t)mat_ncols;
        T sum = 0;
        for (int32_t i = ptr_t[c]; i < ptr_t[c + 1]; i++) sum += std::pow(std::abs(fval_t[i]), alpha);
        return sum;
      }
    }
    return 0;
  }
};

// ProxIndHalfspaceKernel + ProjectHalfspace (src/prox/prox_ind_halfspace.cu:31-86): the layout is
// ALWAYS planar (Vector(count, dim, false, tx, ...), :67-68) whatever `interleaved` says; a is either
// per group (count*dim, planar) or one normal of dim entries (read as Vector(count, dim, true, 0), :79)
template <class T>
void halfspace_run(T* res, const T* arg, size_t count, size_t dim, const T* a, size_t sz_a, const T* b, size_t sz_b) {
  PAR_FOR
  for (ssz tx = 0; tx < (ssz)count; tx++) {
    const T t = (sz_b == count) ? b[tx] : b[0];
    const bool per_group = sz_a == count * dim;
    T sq_norm = 0, iprod = 0;
    for (size_t i = 0; i < dim; i++) {
      const T n = per_group ? a[tx + count * i] : a[i];
      sq_norm += n * n;
      iprod += n * arg[tx + count * i];
    }
    for (size_t i = 0; i < dim; i++) {
      const T n = per_group ? a[tx + count * i] : a[i];
      res[tx + count * i] = arg[tx + count * i] - (std::max(static_cast<T>(0), iprod - t) / sq_norm) * n;
    }
  }
}
// ProxIndSOCKernel (src/prox/prox_ind_soc.cu:30-77): (x_1..x_{dim-1}, y) planar, alpha unused (== 1 enforced at Initialize)
template <class T>
void soc_run(T* res, const T* arg, size_t count, size_t dim) {
  PAR_FOR
  for (ssz tx = 0; tx < (ssz)count; tx++) {
    const T y0 = arg[count * (dim - 1) + tx];
    T norm_x0 = 0;
    for (size_t i = 0; i < dim - 1; i++) norm_x0 += arg[tx + count * i] * arg[tx + count * i];
    norm_x0 = std::sqrt(norm_x0);
    if (norm_x0 <= y0) {
      for (size_t i = 0; i < dim - 1; i++) res[tx + count * i] = arg[tx + count * i];
      res[count * (dim - 1) + tx] = y0;
    } else if (norm_x0 <= -y0) {
      for (size_t i = 0; i < dim - 1; i++) res[tx + count * i] = 0;
      res[count * (dim - 1) + tx] = 0;
    } else {
      const T fac = (y0 + norm_x0) / (2 * norm_x0);
      for (size_t i = 0; i < dim - 1; i++) res[tx + count * i] = fac * arg[tx + count * i];
      res[count * (dim - 1) + tx] = fac * norm_x0;
    }
  }
}
// ProxIndSumKernel (src/prox/prox_ind_sum.cu:30-66); res already holds a copy of arg (:119)
template <class T>
void ind_sum_run(T* res, const T* arg, const T* tau_diag, const size_t* inds, size_t count, size_t dim, T total_sum, T tau, bool inv) {
  for (size_t tx = 0; tx < count; tx++) {       // serial: groups may share indices, the last writer wins as on one stream
    T sum_arg = 0, sum_tau = 0;
    for (size_t i = 0; i < dim; i++) {
      T mytau = tau_diag[inds[tx * dim + i]] * tau;
      if (inv) mytau = (T)(1. / (double)mytau);
      sum_arg += arg[inds[tx * dim + i]];
      sum_tau += mytau;
    }
    for (size_t i = 0; i < dim; i++) {
      T mytau = tau_diag[inds[tx * dim + i]] * tau;
      if (inv) mytau = (T)(1. / (double)mytau);
      res[inds[tx * dim + i]] = arg[inds[tx * dim + i]] - mytau * (sum_arg - total_sum) / sum_tau;
    }
  }
}
// ElemOperationIndSum (include/prost/prox/elemop/elem_operation_ind_sum.hpp:41-60)
template <class T>
void elem_ind_sum_run(T* res, const T* arg, size_t count, size_t dim, bool interleaved) {
  PAR_FOR
  for (ssz t = 0; t < (ssz)count; t++) {
    View v{count, dim, interleaved, (size_t)t};
    T tl = 0;
    for (size_t i = 0; i < dim; i++) tl += arg[v.at(i)];
    tl = (T)(((double)tl - 1.) / (double)static_cast<T>(dim));
    for (size_t i = 0; i < dim; i++) res[v.at(i)] = arg[v.at(i)] - tl;
  }
}

// ElemOperationIndSimplex (include/prost/prox/elemop/elem_operation_ind_simplex.hpp:40-119): projection onto
// the unit simplex by sorting (ShellSort :97-115, descending) and the threshold search of arXiv:1101.6081
template <class T>
void elem_ind_simplex_run(T* res, const T* arg, size_t count, size_t dim, bool interleaved) {
  if (dim > 1024) throw OrcError("ind_simplex: dim exceeds MAX_DIM = 1024 (elem_operation_ind_simplex.hpp:27)");
  PAR_FOR
  for (ssz t = 0; t < (ssz)count; t++) {
    View v{count, dim, interleaved, (size_t)t};
    std::vector<T> a(dim);
    for (size_t i = 0; i < dim; i++) a[i] = arg[v.at(i)];
    const int gaps[6] = {132, 57, 23, 10, 4, 1};
    for (int k = 0; k < 6; k++) {
      const int gap = gaps[k];
      for (int i = gap; i < (int)dim; i++) {
        const T temp = a[i];
        int j = i;
        for (; (j >= gap) && (a[j - gap] <= temp); j -= gap) a[j] = a[j - gap];
        a[j] = temp;
      }
    }
    bool bget = false;
    T tmpsum = 0, tmax = 0;
    for (int ii = 1; ii <= (int)dim - 1; ii++) {
      tmpsum += a[ii - 1];
      tmax = (T)(((double)tmpsum - 1.) / (double)(T)ii);
      if (tmax >= a[ii]) { bget = true; break; }
    }
    if (!bget) tmax = (T)(((double)(T)(tmpsum + a[dim - 1]) - 1.0) / (double)(T)dim);
    for (size_t i = 0; i < dim; i++) res[v.at(i)] = std::max(arg[v.at(i)] - tmax, static_cast<T>(0));
  }
}

// ---------------------------------------------------------------------------------
// Prox tree
// ---------------------------------------------------------------------------------
enum { PK_ELEM, PK_MOREAU, PK_ZERO, PK_EPI_QUAD, PK_TRANSFORM, PK_PERMUTE, PK_HALFSPACE, PK_SOC, PK_IND_SUM };
enum { ORC_OP_IND_SUM = 2, ORC_OP_IND_SIMPLEX = 3 };   // elem_operation:ind_sum / :ind_simplex (no coefficients)

template <class T>
struct Prox {
  int kind; size_t index, size; bool diagsteps;
  size_t count = 0, dim = 0; bool interleaved = false; int op = 0, fn = 0;
  std::array<std::vector<T>, 7> coeffs;
  std::vector<T> a, b, c;
  std::unique_ptr<Prox<T>> child;
  std::vector<T> scaled_arg, scaled_tau;
  std::vector<int> perm;                        // ProxPermute
  std::vector<size_t> inds, inds2; size_t count2 = 0, dim2 = 0; T sum = 0, sum2 = 0; bool two = false;   // ProxIndSum
  T alpha = 1;                                  // ProxIndSOC

  void rehome() {
    for (auto& cv : coeffs) numa_rehome(cv);
    numa_rehome(a); numa_rehome(b); numa_rehome(c); numa_rehome(scaled_arg); numa_rehome(scaled_tau);
    if (child) child->rehome();
  }
  // Prox::Initialize chain (prox_moreau.cu:73-87, prox_ind_epi_quad.cu:137-169)
  void initialize() {
    if (kind == PK_MOREAU) { scaled_arg.assign(size, 0); child->initialize(); }
    if (kind == PK_TRANSFORM) {                                       // prox_transform.cu:113-141
      for (T& v : coeffs[0]) if (v == 0) throw OrcError("ProxTransform: Vector 'a' isn't allowed to contain zero element. (Division by zero)");
      scaled_arg.assign(size, 0); scaled_tau.assign(size, 0);
      child->initialize();
    }
    if (kind == PK_PERMUTE) {                                         // prox_permute.cu:62-93
      scaled_arg.assign(size, 0);
      if (perm.size() != child->size) {
        std::stringstream ss;
        ss << "Permutation vector has wrong size (" << perm.size() << ") instead of " << child->size << ".";
        throw OrcError(ss.str());
      }
      child->initialize();
    }
    if (kind == PK_HALFSPACE) {                                       // prox_ind_halfspace.cu:130-151
      if (a.size() != count * dim && a.size() != dim) throw OrcError("Wrong input: Coefficient a has to have dimension count*dim or dim!");
      if (b.size() != count && b.size() != 1) throw OrcError("Wrong input: Coefficient b has to have dimension count or 1!");
    }
    if (kind == PK_SOC && alpha != 1) throw OrcError("ProxIndSOC: Only alpha = 1 implemented right now.");   // prox_ind_soc.cu:118-122
    if (kind == PK_IND_SUM) {                                         // prox_ind_sum.cu:69-83
      if (count * dim != inds.size()) throw OrcError("ProxIndSum: dimensions dont fit");
      if (two && count2 * dim2 != inds2.size()) throw OrcError("ProxIndSum: dimensions dont fit");
    }
    if (kind == PK_EPI_QUAD) {
      if (a.size() != count && a.size() != 1) throw OrcError("Wrong input: Coefficient a has to have dimension count or 1!");
      for (T& v : a) if (v <= 0) throw OrcError("Wrong input: Coefficient a must be greater 0!");
      if (b.size() != count * (dim - 1)) throw OrcError("Wrong input: Coefficient b has to have dimension count*(dim-1)!");
      if (c.size() != count && c.size() != 1) throw OrcError("Wrong input: Coefficient c has to have dimension count or 1!");
    }
  }
  // EvalLocal: pointers already offset by index
  void eval_local(T* res, const T* arg, const T* tau_diag, T tau, bool invert) {
    switch (kind) {
      case PK_ELEM: {
        if (op == ORC_OP_IND_SUM) { elem_ind_sum_run<T>(res, arg, count, dim, interleaved); break; }
        if (op == ORC_OP_IND_SIMPLEX) { elem_ind_simplex_run<T>(res, arg, count, dim, interleaved); break; }
        const T* cp[7]; T cv[7];
        for (int i = 0; i < 7; i++) {
          if (coeffs[i].size() > 1) { cp[i] = coeffs[i].data(); cv[i] = 0; }     // prox_elem_operation.inl:160-168
          else { cp[i] = nullptr; cv[i] = coeffs[i][0]; }
        }
        prox_elem_run<T>(op, fn, res, arg, tau_diag, tau, invert, count, dim, interleaved, cp, cv);
      } break;
      case PK_ZERO:                                                   // prox_zero.cu:37-48
        if (res != arg) std::memmove(res, arg, size * sizeof(T));
        break;
      case PK_EPI_QUAD:
        epi_quad_run<T>(res, arg, count, dim, a.size() != 1 ? a.data() : nullptr, a[0], b.data(),
                        c.size() != 1 ? c.data() : nullptr, c[0]);
        break;
      case PK_TRANSFORM: {                                            // prox_transform.cu:167-221
        auto co = [&](int k, ssz i) { return coeffs[k].size() > 1 ? coeffs[k][i] : coeffs[k][0]; };
        PAR_FOR
        for (ssz i = 0; i < (ssz)size; i++) {                         // PrescaleArgument :27-52, PrescaleStepSize :54-78
          T tau2 = tau * tau_diag[i];
          if (invert) tau2 = 1 / tau2;
          const T a_ = co(0, i), b_ = co(1, i), c_ = co(2, i), d_ = co(3, i), e_ = co(4, i);
          scaled_arg[i] = (a_ * (arg[i] - tau2 * d_)) / (1 + tau2 * e_) - b_;
          scaled_tau[i] = (a_ * a_ * c_ * tau2) / (1 + tau2 * e_);
        }
        child->eval_local(res, scaled_arg.data(), scaled_tau.data(), 1, false);   // EvalLocal on local ranges: the child's index is not applied
        PAR_FOR
        for (ssz i = 0; i < (ssz)size; i++) res[i] = (res[i] + co(1, i)) / co(0, i);   // Postscale :80-97
      } break;
      case PK_PERMUTE: {                                              // prox_permute.cu:101-143: tau_diag is NOT permuted
        const size_t n = perm.size();
        for (size_t i = 0; i < n; i++) res[i] = arg[perm[i]];
        child->eval_local(scaled_arg.data(), res, tau_diag, tau, invert);
        for (size_t i = 0; i < n; i++) res[perm[i]] = scaled_arg[i];
      } break;
      case PK_HALFSPACE: halfspace_run<T>(res, arg, count, dim, a.data(), a.size(), b.data(), b.size()); break;
      case PK_SOC: soc_run<T>(res, arg, count, dim); break;
      case PK_IND_SUM:
        if (res != arg) std::memmove(res, arg, size * sizeof(T));     // "zero prox on other indices" :119
        ind_sum_run<T>(res, arg, tau_diag, inds.data(), count, dim, sum, tau, invert);
        if (two) ind_sum_run<T>(res, arg, tau_diag, inds2.data(), count2, dim2, sum2, tau, invert);
        break;
      case PK_MOREAU: {                                               // prox_moreau.cu:98-134
        PAR_FOR
        for (ssz i = 0; i < (ssz)size; i++)                           // MoreauPrescale :29-43
          scaled_arg[i] = invert ? arg[i] * (tau * tau_diag[i]) : arg[i] / (tau * tau_diag[i]);
        child->eval_local(res, scaled_arg.data(), tau_diag, tau, !invert);
        PAR_FOR
        for (ssz i = 0; i < (ssz)size; i++) {                         // MoreauPostscale :45-61
          if (invert) res[i] = arg[i] - res[i] / (tau * tau_diag[i]);
          else res[i] = arg[i] - tau * tau_diag[i] * res[i];
        }
      } break;
    }
  }
  void eval(T* res, const T* arg, const T* tau_diag, T tau, bool invert = false) {   // prox.cu:27-43
    eval_local(res + index, arg + index, tau_diag + index, tau, invert);
  }
  // get_separable_structure (prox.cu:74-78, prox_separable_sum.hpp:67-81, prox_moreau.cu:142-147)
  void separable(std::vector<std::tuple<size_t, size_t, size_t>>& sep) const {
    if (kind == PK_MOREAU || kind == PK_TRANSFORM || kind == PK_PERMUTE) { child->separable(sep); return; }
    if (kind == PK_ELEM || kind == PK_EPI_QUAD || kind == PK_HALFSPACE || kind == PK_SOC) {
      if (interleaved) for (size_t i = 0; i < count; i++) sep.emplace_back(index + i * dim, dim, 1);
      else for (size_t i = 0; i < count; i++) sep.emplace_back(index + i, dim, count);
      return;
    }
    sep.emplace_back(index, size, 1);
  }
  size_t end() const { return index + size - 1; }
};

// ---------------------------------------------------------------------------------
// Problem
// ---------------------------------------------------------------------------------
struct ProblemBase {
  int dtype; size_t nrows, ncols;
  virtual ~ProblemBase() {}
};

template <class T>
struct Problem : ProblemBase {
  typedef std::vector<std::shared_ptr<Prox<T>>> ProxList;
  std::vector<Block<T>> blocks;
  ProxList prox_g, prox_f, prox_gstar, prox_fstar;
  int scaling_type = 0;   // 0 alpha, 1 identity, 2 custom
  T scaling_alpha = 1;
  std::vector<T> left, right;   // squared preconditioners Sigma, Tau (problem.hpp:127-131)
  size_t lin_nrows = 0, lin_ncols = 0;
  bool dualized = false;
  bool quirks = true;     // reference quirks (negate<float>, diags adjoint grid)
  bool initialized = false;

  // LinearOperator::Initialize (linearoperator.cu:84-125)
  void linop_initialize() {
    lin_nrows = lin_ncols = 0;
    bool overlap = false;
    for (size_t i = 0; i < blocks.size(); i++) {
      const Block<T>& bi = blocks[i];
      lin_nrows = std::max(bi.row + bi.nrows, lin_nrows);
      lin_ncols = std::max(bi.col + bi.ncols, lin_ncols);
      for (size_t j = i + 1; j < blocks.size(); j++) {
        const Block<T>& bj = blocks[j];
        size_t x1 = bi.col, y1 = bi.row, x2 = bi.col + bi.ncols - 1, y2 = bi.row + bi.nrows - 1;
        size_t a1 = bj.col, b1 = bj.row, a2 = bj.col + bj.ncols - 1, b2 = bj.row + bj.nrows - 1;
        overlap |= (x1 <= a2) && (x2 >= a1) && (y1 <= b2) && (y2 >= b1);
      }
    }
    if (overlap) throw OrcError("Blocks are overlapping inside the linear operator. Recheck the indices.");
  }
  // K (not dualized) forward / adjoint with beta (linearoperator.cu:135-170)
  void K_eval(T* res, size_t nres, const T* rhs, T beta, bool adjoint) const {
    if (beta == 0) std::fill(res, res + nres, (T)0);
    else if (beta != 1) for (size_t i = 0; i < nres; i++) res[i] = beta * res[i];
    for (const auto& b : blocks) {
      if (!adjoint) b.add(res + b.row, rhs + b.col);           // block.cu:47-57
      else b.add_adj(res + b.col, rhs + b.row, quirks);        // block.cu:59-68
    }
  }
  // problem_->linop()->Eval / EvalAdjoint honouring Dualize (dual_linearoperator.cu:39-80)
  void linop_eval(T* res, const T* rhs, T beta = 0) const {
    if (!dualized) { K_eval(res, nrows, rhs, beta, false); return; }
    dual_apply(res, nrows, rhs, beta, true);
  }
  void linop_eval_adjoint(T* res, const T* rhs, T beta = 0) const {
    if (!dualized) { K_eval(res, ncols, rhs, beta, true); return; }
    dual_apply(res, ncols, rhs, beta, false);
  }
  void dual_apply(T* res, size_t nres, const T* rhs, T beta, bool child_adjoint) const {
    if (beta == 0) std::fill(res, res + nres, (T)0);
    else if (beta != 1) for (size_t i = 0; i < nres; i++) res[i] = -beta * res[i];
    for (const auto& b : blocks) {
      if (child_adjoint) b.add_adj(res + b.col, rhs + b.row, quirks);
      else b.add(res + b.row, rhs + b.col);
    }
    // thrust::negate<float> even for T=double (dual_linearoperator.cu:56-57,:78-79)
    if (quirks) for (size_t i = 0; i < nres; i++) res[i] = (T)(-(float)res[i]);
    else for (size_t i = 0; i < nres; i++) res[i] = -res[i];
  }
  T K_row_sum(size_t r, T alpha) const {                         // linearoperator.cu:222-236
    T sum = 0;
    for (const auto& b : blocks) { if (r < b.row || r >= b.row + b.nrows) continue; sum += b.row_sum(r - b.row, alpha); }
    return sum;
  }
  T K_col_sum(size_t c, T alpha) const {                         // linearoperator.cu:238-252
    T sum = 0;
    for (const auto& b : blocks) { if (c < b.col || c >= b.col + b.ncols) continue; sum += b.col_sum(c - b.col, alpha); }
    return sum;
  }

  static void sort_by_index(ProxList& l) {
    std::sort(l.begin(), l.end(), [](const std::shared_ptr<Prox<T>>& a, const std::shared_ptr<Prox<T>>& b) { return a->index < b->index; });
  }
  // AddZeroProx (problem.cu:93-158)
  static void add_zero_prox(ProxList& proxs, size_t n, const std::string& name) {
    size_t num = proxs.size();
    if (num == 0) return;
    ProxList s = proxs; sort_by_index(s);
    auto zero = [](size_t idx, size_t size) {
      auto p = std::make_shared<Prox<T>>(); p->kind = PK_ZERO; p->index = idx; p->size = size; p->diagsteps = true; return p;
    };
    if (s[0]->index > 0) proxs.push_back(zero(0, s[0]->index));
    for (size_t i = 0; i + 1 < num; i++)
      if (s[i]->end() < (s[i + 1]->index - 1))
        proxs.push_back(zero(s[i]->end() + 1, s[i + 1]->index - s[i]->end() - 1));
    if (s[num - 1]->end() != (n - 1)) {
      if (s[num - 1]->end() < (n - 1)) proxs.push_back(zero(s[num - 1]->end() + 1, (n - 1) - s[num - 1]->end()));
      else {
        std::stringstream ss;
        ss << name << " (AddZeroProx): Last prox operator ends after the domain: [" << s[num - 1]->index << ", "
           << s[num - 1]->end() << "], end = " << n - 1 << "." << std::endl;
        throw OrcError(ss.str());
      }
    }
  }
  // CheckDomainProx (problem.cu:48-89)
  static void check_domain(const ProxList& proxs, size_t n, const std::string& name) {
    size_t num = proxs.size();
    if (num == 0) return;
    ProxList s = proxs; sort_by_index(s);
    for (size_t i = 0; i + 1 < num; i++)
      if (s[i]->end() != (s[i + 1]->index - 1)) {
        std::stringstream ss;
        ss << name << " (CheckDomainProx): Prox operators are overlapping: [" << s[i]->index << ", " << s[i]->end()
           << "] and [" << s[i + 1]->index << ", " << s[i + 1]->end() << "]." << std::endl;
        throw OrcError(ss.str());
      }
    if (s[num - 1]->end() != (n - 1)) {
      std::stringstream ss;
      if (s[num - 1]->end() < (n - 1))
        ss << name << " (CheckDomainProx): Last prox operator ends too early: [";
      else
        ss << name << " (CheckDomainProx): Last prox operator ends after the domain: [";
      ss << s[num - 1]->index << ", " << s[num - 1]->end() << "], end = " << n - 1 << "." << std::endl;
      throw OrcError(ss.str());
    }
  }
  // AveragePreconditioners (problem.cu:503-536)
  static void average(std::vector<T>& precond, const ProxList& prox) {
    std::vector<std::tuple<size_t, size_t, size_t>> ics;
    for (auto& p : prox) if (!p->diagsteps) p->separable(ics);
    for (auto& t : ics) {
      size_t idx = std::get<0>(t), cnt = std::get<1>(t), sd = std::get<2>(t);
      T avg = 0;
      for (size_t c = 0; c < cnt; c++) avg += precond[idx + c * sd];
      avg /= static_cast<T>(cnt);
      for (size_t c = 0; c < cnt; c++) precond[idx + c * sd] = avg;
    }
  }
  // Problem::Initialize (problem.cu:196-323)
  void initialize() {
    linop_initialize();
    if (prox_f.empty() && prox_fstar.empty()) throw OrcError("No proximal operator for f or fstar specified.");
    if (prox_g.empty() && prox_gstar.empty()) throw OrcError("No proximal operator for g or gstar specified.");
    if (!prox_f.empty() && !prox_fstar.empty()) throw OrcError("Proximal operator for f AND fstar specified. Only set one!");
    if (!prox_g.empty() && !prox_gstar.empty()) throw OrcError("Proximal operator for g AND gstar specified. Only set one!");
    if (!prox_f.empty()) add_zero_prox(prox_f, nrows, "prox_f");
    if (!prox_g.empty()) add_zero_prox(prox_g, ncols, "prox_g");
    if (!prox_fstar.empty()) add_zero_prox(prox_fstar, nrows, "prox_fstar");
    if (!prox_gstar.empty()) add_zero_prox(prox_gstar, ncols, "prox_gstar");
    check_domain(prox_g, ncols, "prox_g");
    check_domain(prox_f, nrows, "prox_f");
    check_domain(prox_gstar, ncols, "prox_gstar");
    check_domain(prox_fstar, nrows, "prox_fstar");
    for (auto& p : prox_f) p->initialize();
    for (auto& p : prox_fstar) p->initialize();
    for (auto& p : prox_g) p->initialize();
    for (auto& p : prox_gstar) p->initialize();
    if (scaling_type == 0) {
      left.assign(nrows, 0); right.assign(ncols, 0);
      T value = 1;                                      // carried across rows AND into the column loop (:262-287)
      for (size_t r = 0; r < nrows; r++) {
        T rowsum = K_row_sum(r, scaling_alpha);
        if (rowsum > 0) value = (T)(1. / (double)rowsum);
        left[r] = value;
      }
      for (size_t c = 0; c < ncols; c++) {
        T colsum = K_col_sum(c, (T)(2. - (double)scaling_alpha));
        if (colsum > 0) value = (T)(1. / (double)colsum);
        right[c] = value;
      }
    } else if (scaling_type == 1) {
      left.assign(nrows, 1); right.assign(ncols, 1);
    } else {
      if (left.size() != nrows || right.size() != ncols)
        throw OrcError("Preconditioners/diagonal scaling vectors do not fit the size of linear operator.");
    }
    average(right, prox_g.empty() ? prox_gstar : prox_g);
    average(left, prox_f.empty() ? prox_fstar : prox_f);
    initialized = true;
  }
  // Problem::Dualize (problem.cu:539-547)
  void dualize() {
    prox_g.swap(prox_fstar);
    prox_gstar.swap(prox_f);
    std::swap(nrows, ncols);
    dualized = !dualized;
    std::swap(left, right);
  }
  // Problem::normest (problem.cu:429-500)
  T normest(T tol, int max_iters) {
    const size_t n = ncols, m = nrows;
    std::vector<T> x(n), Ax(m), x_temp(n), Ax_temp(m);
    GlibcRand rng(1);
    for (size_t i = 0; i < n; i++) x[i] = (T)rng.next() / (T)RAND_MAX;
    T norm = 0, norm_prev;
    for (int it = 0; it < max_iters; it++) {
      norm_prev = norm;
      for (size_t i = 0; i < n; i++) x_temp[i] = std::sqrt(right[i]) * x[i];
      linop_eval(Ax_temp.data(), x_temp.data());
      for (size_t i = 0; i < m; i++) Ax[i] = std::sqrt(left[i]) * Ax_temp[i];
      // (the reference reduces in T with thrust, order unspecified on the device: ExactSum, then narrowed to T)
      T norm_Ax = (T)std::sqrt(exact_sum_sq(Ax.data(), m));
      for (size_t i = 0; i < m; i++) Ax_temp[i] = std::sqrt(left[i]) * Ax[i];
      linop_eval_adjoint(x_temp.data(), Ax_temp.data());
      for (size_t i = 0; i < n; i++) x[i] = std::sqrt(right[i]) * x_temp[i];
      T norm_x = (T)std::sqrt(exact_sum_sq(x.data(), n));
      norm = norm_x / norm_Ax;
      if ((double)std::abs(norm_prev - norm) < (double)(tol * norm)) break;
      for (size_t i = 0; i < n; i++) x[i] = x[i] / norm_x;
    }
    return norm;
  }
};

// ---------------------------------------------------------------------------------
// Solver + backends
// ---------------------------------------------------------------------------------
struct SolverBase {
  virtual ~SolverBase() {}
  virtual void initialize() = 0;
  virtual void iterate(int iters) = 0;
  virtual void rehome() = 0;          // timing runs: re-place the pages of every large vector for the current thread team
  virtual void solve(int* result, int* iters_done) = 0;
  virtual void get(double* x, double* z, double* y, double* w) = 0;
  virtual void scalars(double* out) = 0;
  orc_interm_cb interm_cb = nullptr; orc_stop_cb stop_cb = nullptr; void* cb_user = nullptr;
  orc_allreduce_cb allreduce = nullptr; void* ar_user = nullptr; size_t g_nrows = 0, g_ncols = 0;
};

template <class T>
struct Solver : SolverBase {
  Problem<T>* prob;
  bool is_admm;
  orc_pdhg_opts po; orc_admm_opts ao;
  // Solver<T>::Options (solver.hpp:39-70), values narrowed to T like factory.cpp:992-1012
  T tol_rel_primal, tol_rel_dual, tol_abs_primal, tol_abs_dual;
  int max_iters, num_cback_calls; bool verbose, solve_dual;
  std::vector<T> x0, y0;
  // Backend state
  T primal_res = 0, dual_res = 0, primal_var_norm = 0, dual_var_norm = 0;
  size_t iteration = 0; int arb_l = 0, arb_u = 0;
  typename Problem<T>::ProxList pg, pfs;    // PDHG: prox_g, prox_fstar ; ADMM: prox_g, prox_f
  // PDHG (backend_pdhg.hpp:106-151)
  std::vector<T> x, y, x_prev, y_prev, temp, kx, kty, kx_prev, kty_prev;
  T tau = 0, sigma = 0, theta = 1, arg_alpha = 0;
  // ADMM (backend_admm.hpp)
  std::vector<T> x_half, z_half, x_proj, z_proj, x_dual, z_dual, temp1, temp2, temp3;
  T rho = 0, delta = 0;

  size_t eps_rows() const { return g_nrows ? g_nrows : prob->nrows; }
  size_t eps_cols() const { return g_ncols ? g_ncols : prob->ncols; }
  // Backend::eps_primal / eps_dual (backend.hpp:71-74)
  T eps_primal() const { return (T)(std::sqrt((double)eps_rows()) * (double)tol_abs_primal + (double)(tol_rel_primal * primal_var_norm)); }
  T eps_dual() const { return (T)(std::sqrt((double)eps_cols()) * (double)tol_abs_dual + (double)(tol_rel_dual * dual_var_norm)); }

  static std::shared_ptr<Prox<T>> wrap_moreau(const std::shared_ptr<Prox<T>>& p) {
    // ProxMoreau(conjugate): copies index/size/diagsteps (prox_moreau.cu:63-67); shares the child
    auto m = std::make_shared<Prox<T>>();
    m->kind = PK_MOREAU; m->index = p->index; m->size = p->size; m->diagsteps = p->diagsteps;
    m->child.reset(new Prox<T>(clone(*p)));
    m->initialize();
    return m;
  }
  static Prox<T> clone(const Prox<T>& p) {
    Prox<T> q; q.kind = p.kind; q.index = p.index; q.size = p.size; q.diagsteps = p.diagsteps;
    q.count = p.count; q.dim = p.dim; q.interleaved = p.interleaved; q.op = p.op; q.fn = p.fn;
    q.coeffs = p.coeffs; q.a = p.a; q.b = p.b; q.c = p.c; q.scaled_arg = p.scaled_arg; q.scaled_tau = p.scaled_tau;
    q.perm = p.perm; q.inds = p.inds; q.inds2 = p.inds2; q.count2 = p.count2; q.dim2 = p.dim2; q.sum = p.sum; q.sum2 = p.sum2;
    q.two = p.two; q.alpha = p.alpha;
    if (p.child) q.child.reset(new Prox<T>(clone(*p.child)));
    return q;
  }

  // Solver::Initialize (solver.cu:68-120)
  void initialize() override {
    try { prob->initialize(); }
    catch (const std::exception& e) { throw OrcError(std::string("Failed to initialize the problem. Reason: ") + e.what()); }
    if (solve_dual) { prob->dualize(); x0.swap(y0); }
    try { if (is_admm) admm_initialize(); else pdhg_initialize(); }
    catch (const std::exception& e) { throw OrcError(std::string("Failed to initialize the backend. Reason: ") + e.what()); }
  }

  // ---------------- PDHG: BackendPDHG::Initialize (backend_pdhg.cu:201-309) ----------------
  void pdhg_initialize() {
    size_t m = prob->nrows, n = prob->ncols, l = std::max(m, n);
    x.assign(n, 0); x_prev.assign(n, 0); kty_prev.assign(n, 0); kty.assign(n, 0);
    y.assign(m, 0); y_prev.assign(m, 0); kx.assign(m, 0); kx_prev.assign(m, 0); temp.assign(l, 0);
    iteration = 0; tau = (T)po.tau0; sigma = (T)po.sigma0; theta = 1;
    arb_l = arb_u = 0; arg_alpha = (T)po.arg_alpha0;
    pg.clear(); pfs.clear();
    if (prob->prox_g.empty()) {
      if (prob->prox_gstar.empty()) throw OrcError("Neither prox_g nor prox_gstar specified.");
      for (auto& p : prob->prox_gstar) pg.push_back(wrap_moreau(p));
    } else pg = prob->prox_g;
    if (prob->prox_fstar.empty()) {
      if (prob->prox_f.empty()) throw OrcError("Neither prox_f nor prox_fstar specified.");
      for (auto& p : prob->prox_f) pfs.push_back(wrap_moreau(p));
    } else pfs = prob->prox_fstar;
    primal_var_norm = dual_var_norm = primal_res = dual_res = 0;
    if (po.scale_steps_operator) {
      T norm = prob->normest((T)1e-6, 100);
      if ((double)std::abs(norm - 1) > 0.1) { tau /= norm; sigma /= norm; }
    }
    if (!x0.empty()) { if (x0.size() == n) { x = x0; x_prev = x0; } else throw OrcError("Initial primal solution has wrong size."); }
    if (!y0.empty()) { if (y0.size() == m) { y = y0; y_prev = y0; } else throw OrcError("Initial dual solution has wrong size."); }
    rehome_problem();
    for (auto* v : {&x, &y, &x_prev, &y_prev, &temp, &kx, &kty, &kx_prev, &kty_prev}) numa_rehome(*v);
  }
  void rehome() override {
    rehome_problem();
    for (auto* v : {&x, &y, &x_prev, &y_prev, &temp, &kx, &kty, &kx_prev, &kty_prev, &x_half, &z_half, &x_proj, &z_proj, &x_dual, &z_dual, &temp1, &temp2, &temp3}) numa_rehome(*v);
  }
  // multi-threaded timing runs: every large vector first-touched by the thread that streams it (numa_rehome); no effect on results
  void rehome_problem() {
    if (g_threads <= 1) return;
    numa_rehome(prob->left); numa_rehome(prob->right);
    for (auto& b : prob->blocks) b.rehome();
    for (auto* l : {&prob->prox_g, &prob->prox_f, &prob->prox_gstar, &prob->prox_fstar, &pg, &pfs}) for (auto& pr : *l) pr->rehome();
  }

  // BackendPDHG::PerformIteration (backend_pdhg.cu:313-381)
  void pdhg_iteration() {
    const ssz n = (ssz)prob->ncols, m = (ssz)prob->nrows;
    const T* Tr = prob->right.data(); const T* Sl = prob->left.data();
    { T* tp = temp.data(); const T* xp = x.data(); const T* kp = kty.data(); const T t = tau;
      PAR_FOR
      for (ssz i = 0; i < n; i++) tp[i] = xp[i] - t * Tr[i] * kp[i]; }            // :38-51
    x.swap(x_prev);
    for (auto& p : pg) p->eval(x.data(), temp.data(), Tr, tau);
    kx.swap(kx_prev);
    prob->linop_eval(kx.data(), x.data());
    { T* tp = temp.data(); const T* yp = y.data(); const T* a = kx.data(); const T* b = kx_prev.data();
      const T s = sigma, th = theta;
      PAR_FOR
      for (ssz i = 0; i < m; i++) tp[i] = yp[i] + s * Sl[i] * ((1 + th) * a[i] - th * b[i]); }   // :54-70
    y.swap(y_prev);
    for (auto& p : pfs) p->eval(y.data(), temp.data(), Sl, sigma);
    pdhg_update_residuals_and_stepsizes();
    iteration++;
    kty.swap(kty_prev);
    prob->linop_eval_adjoint(kty.data(), y.data());
  }

  // thrust::transform_reduce with tuple-sum: sequential left fold in T on the host
  // backend; an OpenMP reduction (double) when the oracle runs multi-threaded.
  template <class F> void reduce2(ssz n, F f, T& s0, T& s1) const {
    if (g_threads <= 1) {
      T a = 0, b = 0;
      for (ssz i = 0; i < n; i++) { T d, w; f(i, d, w); a = a + d; b = b + w; }
      s0 = a; s1 = b;
    } else {
      double a = 0, b = 0;
#pragma omp parallel for schedule(static) num_threads(g_threads) reduction(+ : a, b)
      for (ssz i = 0; i < n; i++) { T d, w; f(i, d, w); a += (double)d; b += (double)w; }
      s0 = (T)a; s1 = (T)b;
    }
  }

  // BackendPDHG::UpdateResidualsAndStepsizes (backend_pdhg.cu:385-489)
  void pdhg_update_residuals_and_stepsizes() {
    if (iteration == 0 || (iteration % (size_t)po.residual_iter) == 0) {
      const T* Sl = prob->left.data(); const T* Tr = prob->right.data();
      const T sg = sigma, th = theta, ta = tau;
      T p0, p1, d0, d1;
      { const T* a0 = y_prev.data(); const T* a1 = y.data(); const T* a3 = kx_prev.data(); const T* a4 = kx.data();
        reduce2((ssz)prob->nrows, [&](ssz i, T& dd, T& ww) {                       // primal_residual_transform :97-120
          const T sd = Sl[i];
          const T z_hat = (a0[i] - a1[i]) / (sg * std::sqrt(sd)) + std::sqrt(sd) * ((1 + th) * a4[i] - th * a3[i]);
          const T diff = z_hat - std::sqrt(sd) * a4[i];
          dd = diff * diff; ww = z_hat * z_hat; }, p0, p1); }
      { const T* a0 = x_prev.data(); const T* a1 = x.data(); const T* a3 = kty_prev.data(); const T* a4 = kty.data();
        reduce2((ssz)prob->ncols, [&](ssz i, T& dd, T& ww) {                       // dual_residual_transform :73-94
          const T td = Tr[i];
          const T w_hat = (a0[i] - a1[i]) / (ta * std::sqrt(td)) - std::sqrt(td) * a3[i];
          const T diff = w_hat + std::sqrt(td) * a4[i];
          dd = diff * diff; ww = w_hat * w_hat; }, d0, d1); }
      if (allreduce) {
        double v[4] = {(double)p0, (double)p1, (double)d0, (double)d1};
        allreduce(ar_user, v);
        p0 = (T)v[0]; p1 = (T)v[1]; d0 = (T)v[2]; d1 = (T)v[3];
      }
      primal_res = std::sqrt(p0); primal_var_norm = std::sqrt(p1);
      dual_res = std::sqrt(d0); dual_var_norm = std::sqrt(d1);
      T eps_p = eps_primal(), eps_d = eps_dual();
      if (po.stepsize == ORC_STEP_GOLDSTEIN) {                                    // :443-460
        const T arg_delta = (T)po.arg_delta, arg_nu = (T)po.arg_nu;
        T scale = eps_d / eps_p;
        if (dual_res > (scale * primal_res * arg_delta)) {
          tau = tau / (1 - arg_alpha); sigma = sigma * (1 - arg_alpha); arg_alpha = arg_alpha * arg_nu;
        }
        if (dual_res < (scale * primal_res / arg_delta)) {
          tau = tau * (1 - arg_alpha); sigma = sigma / (1 - arg_alpha); arg_alpha = arg_alpha * arg_nu;
        }
      } else if (po.stepsize == ORC_STEP_BOYD) {                                  // :462-476
        const T arb_delta = (T)po.arb_delta, arb_tau = (T)po.arb_tau;
        if ((dual_res < eps_d) && (arb_tau * iteration > arb_l)) {
          tau /= arb_delta; sigma *= arb_delta; arb_u = (int)iteration;
        } else if ((primal_res < eps_p) && (arb_tau * iteration > arb_u)) {
          tau *= arb_delta; sigma /= arb_delta; arb_l = (int)iteration;
        }
      }
    }
    if (po.stepsize == ORC_STEP_ALG2) {                                           // :483-488
      theta = (T)(1. / std::sqrt(1. + 2. * (double)(T)po.alg2_gamma * (double)tau));
      tau = theta * tau;
      sigma = sigma / theta;
    }
  }

  // BackendPDHG::current_solution (backend_pdhg.cu:515-563)
  void pdhg_current_solution(std::vector<T>& px, std::vector<T>& pz, std::vector<T>& dy, std::vector<T>& dw) {
    const size_t n = prob->ncols, m = prob->nrows;
    px = x; dy = y;
    const T* Tr = prob->right.data(); const T* Sl = prob->left.data();
    for (size_t i = 0; i < n; i++) temp[i] = (x_prev[i] - x[i]) / (Tr[i] * tau) - kty_prev[i];                    // :154-160
    dw.assign(temp.begin(), temp.begin() + n);
    for (size_t i = 0; i < m; i++) temp[i] = (y_prev[i] - y[i]) / (sigma * Sl[i]) + (1 + theta) * kx[i] - theta * kx_prev[i];  // :178-185
    pz.assign(temp.begin(), temp.begin() + m);
  }

  // ---------------- ADMM: BackendADMM::Initialize (backend_admm.cu:286-352) ----------------
  void admm_initialize() {
    size_t m = prob->nrows, n = prob->ncols, l = std::max(m, n);
    x_half.assign(n, 0); x_proj.assign(n, 0); x_dual.assign(n, 0);
    z_half.assign(m, 0); z_proj.assign(m, 0); z_dual.assign(m, 0);
    temp1.assign(n, 0); temp2.assign(l, 0); temp3.assign(l, 0);   // temp2 over-allocated: :586-590 writes n entries
    pg.clear(); pfs.clear();
    if (prob->prox_g.empty()) {
      if (prob->prox_gstar.empty()) throw OrcError("Neither prox_g nor prox_gstar specified.");
      for (auto& p : prob->prox_gstar) pg.push_back(wrap_moreau(p));
    } else pg = prob->prox_g;
    if (prob->prox_f.empty()) {
      if (prob->prox_fstar.empty()) throw OrcError("Neither prox_f nor prox_fstar specified.");
      for (auto& p : prob->prox_fstar) pfs.push_back(wrap_moreau(p));
    } else pfs = prob->prox_f;
    delta = (T)ao.arb_delta; rho = (T)ao.rho0; iteration = 0; arb_u = arb_l = 0;
    // the reference leaves the residual members uninitialised here;
    // the oracle zero-initialises them (members default to 0).
    rehome_problem();
    for (auto* v : {&x_half, &z_half, &x_proj, &z_proj, &x_dual, &z_dual, &temp1, &temp2, &temp3}) numa_rehome(*v);
  }

  // GemvPrecondK (backend_admm.cu:199-272): y := alpha*op(S^1/2 K T^1/2) x + beta*y
  void gemv(char op, const T alpha, const std::vector<T>& xv, const T beta, std::vector<T>& yv) {
    const T* Sl = prob->left.data(); const T* Tr = prob->right.data();
    const size_t m = prob->nrows, n = prob->ncols;
    if (op == 'n') {
      for (size_t i = 0; i < n; i++) temp3[i] = std::sqrt(Tr[i]) * xv[i];
      for (size_t i = 0; i < m; i++) yv[i] = (beta / (alpha * std::sqrt(Sl[i]))) * yv[i];
      prob->linop_eval(yv.data(), temp3.data(), 1);
      for (size_t i = 0; i < m; i++) yv[i] = alpha * std::sqrt(Sl[i]) * yv[i];
    } else {
      for (size_t i = 0; i < m; i++) temp3[i] = std::sqrt(Sl[i]) * xv[i];
      for (size_t i = 0; i < n; i++) yv[i] = (beta / (alpha * std::sqrt(Tr[i]))) * yv[i];
      prob->linop_eval_adjoint(yv.data(), temp3.data(), 1);
      for (size_t i = 0; i < n; i++) yv[i] = alpha * std::sqrt(Tr[i]) * yv[i];
    }
  }
  // cgls nrm2: thrust transform_reduce in double (cgls.hpp:152-170)
  static double nrm2d(const std::vector<T>& v, size_t n) { return std::sqrt(exact_sum_sq(v.data(), n)); }     // order-independent: ExactSum
  // cublas<t>nrm2 stand-in for the ADMM residuals (backend_admm.cu:40-50): cuBLAS is not
  // under /root/reference; restated as sqrt of the sum of squares accumulated in double,
  // narrowed to T for T=float as the reference does (result_float).
  static double blas_nrm2(const std::vector<T>& v, size_t n) { return (double)(T)std::sqrt(exact_sum_sq(v.data(), n)); }
  // cgls::Solve (include/prost/cgls.hpp:222-371)
  int cgls_solve(int m, int n, const std::vector<T>& b, std::vector<T>& xs, double shift, double tol, int maxit,
                 std::vector<T>& p, std::vector<T>& q, std::vector<T>& r, std::vector<T>& s, int& iterations) {
    double gamma, normp, normq, norms, norms0, normx, xmax;
    int k = 0, flag = 0, indefinite = 0;
    const T kNegOne = (T)-1., kZero = (T)0., kOne = (T)1., kNegShift = (T)(-shift);
    const double kEps = std::numeric_limits<T>::epsilon();
    std::copy(b.begin(), b.begin() + m, r.begin());
    std::copy(xs.begin(), xs.begin() + n, s.begin());
    normx = nrm2d(xs, n);
    if (normx > 0.) gemv('n', kNegOne, xs, kOne, r);
    gemv('t', kOne, r, kNegShift, s);
    std::copy(s.begin(), s.begin() + n, p.begin());
    norms = nrm2d(s, n);
    norms0 = norms; gamma = norms0 * norms0;
    normx = nrm2d(xs, n); xmax = normx;
    if (norms < kEps) flag = 1;
    for (k = 0; k < maxit && !flag; ++k) {
      gemv('n', kOne, p, kZero, q);
      normp = nrm2d(p, n); normq = nrm2d(q, m);
      double dlt = normq * normq + shift * normp * normp;
      if (dlt <= 0.) indefinite = 1;
      if (dlt == 0.) dlt = kEps;
      T alpha = (T)(gamma / dlt), neg_alpha = (T)(-gamma / dlt);
      for (int i = 0; i < n; i++) xs[i] = alpha * p[i] + xs[i];        // axpy
      for (int i = 0; i < m; i++) r[i] = neg_alpha * q[i] + r[i];
      std::copy(xs.begin(), xs.begin() + n, s.begin());
      gemv('t', kOne, r, kNegShift, s);
      norms = nrm2d(s, n);
      double gamma1 = gamma; gamma = norms * norms;
      T beta = (T)(gamma / gamma1);
      for (int i = 0; i < n; i++) s[i] = beta * p[i] + s[i];
      std::copy(s.begin(), s.begin() + n, p.begin());
      normx = nrm2d(xs, n);
      xmax = std::max(xmax, normx);
      bool converged = (norms <= norms0 * tol) || (normx * tol >= 1.);
      if (converged) break;
    }
    double shrink = normx / xmax;
    if (k == maxit) flag = 2;
    else if (indefinite) flag = 3;
    else if (shrink * shrink <= tol) flag = 4;
    iterations = k;
    return flag;
  }

  // BackendADMM::PerformIteration (backend_admm.cu:355-665)
  void admm_iteration() {
    const size_t m = prob->nrows, n = prob->ncols;
    const T* Sl = prob->left.data(); const T* Tr = prob->right.data();
    const T al = (T)ao.alpha;
    for (size_t i = 0; i < n; i++) temp1[i] = (al * x_half[i] + (1 - al) * x_proj[i] + x_dual[i]) / std::sqrt(Tr[i]);   // :53-67
    for (size_t i = 0; i < m; i++) temp2[i] = std::sqrt(Sl[i]) * (z_half[i] + z_dual[i]);                              // :70-81
    std::copy(temp2.begin(), temp2.begin() + m, z_dual.begin());
    std::copy(temp3.begin(), temp3.begin() + n, x_proj.begin());
    gemv('n', (T)-1, temp1, (T)1, z_dual);
    double cg_tol = ao.cg_tol_min / std::pow((double)static_cast<T>(iteration + 1), ao.cg_tol_pow);   // :408-410
    cg_tol = std::max(cg_tol, ao.cg_tol_max);
    int& taken = last_cg_iters;
    cgls_solve((int)m, (int)n, z_dual, x_proj, 1, cg_tol, ao.cg_max_iter, x_half, z_half, z_proj, x_dual, taken);
    std::copy(x_proj.begin(), x_proj.end(), temp3.begin());
    for (size_t i = 0; i < n; i++) x_proj[i] = std::sqrt(Tr[i]) * (x_proj[i] + temp1[i]);     // x_proj_functor :96-105
    prob->linop_eval(z_proj.data(), x_proj.data());
    for (size_t i = 0; i < n; i++) x_dual[i] = temp1[i] * std::sqrt(Tr[i]) - x_proj[i];       // :108-117
    for (size_t i = 0; i < m; i++) z_dual[i] = temp2[i] / std::sqrt(Sl[i]) - z_proj[i];       // :120-129
    for (size_t i = 0; i < n; i++) temp1[i] = x_proj[i] - x_dual[i];
    for (auto& p : pg) p->eval(x_half.data(), temp1.data(), Tr, 1 / rho);
    for (size_t i = 0; i < m; i++) temp2[i] = z_proj[i] - z_dual[i];
    for (auto& p : pfs) p->eval(z_half.data(), temp2.data(), Sl, rho, true);
    iteration++;
    if (iteration == 0 || (iteration % (size_t)ao.residual_iter) == 0) {                       // :535-663
      double pr, pv, dr, dv;
      std::copy(z_half.begin(), z_half.end(), temp2.begin());
      prob->linop_eval(temp2.data(), x_half.data(), (T)-1);
      for (size_t i = 0; i < m; i++) temp2[i] = std::sqrt(Sl[i]) * temp2[i];
      pr = blas_nrm2(temp2, m);
      for (size_t i = 0; i < m; i++) temp2[i] = std::sqrt(Sl[i]) * z_half[i];
      pv = blas_nrm2(temp2, m);
      for (size_t i = 0; i < n; i++) temp1[i] = -rho * std::pow(Tr[i], (T)-1) * (x_half[i] - x_proj[i] + x_dual[i]);   // get_dual_functor :181-196
      for (size_t i = 0; i < n; i++) temp2[i] = std::sqrt(Tr[i]) * temp1[i];   // NB: writes n entries of the m-vector (reference :586-590)
      dv = blas_nrm2(temp2, n);
      for (size_t i = 0; i < m; i++) temp2[i] = -rho * std::pow(Sl[i], (T)1) * (z_half[i] - z_proj[i] + z_dual[i]);
      prob->linop_eval_adjoint(temp1.data(), temp2.data(), 1);
      for (size_t i = 0; i < n; i++) temp1[i] = std::sqrt(Tr[i]) * temp1[i];
      dr = blas_nrm2(temp1, n);
      if (allreduce) {
        double v[4] = {pr * pr, pv * pv, dr * dr, dv * dv};
        allreduce(ar_user, v);
        pr = std::sqrt(v[0]); pv = std::sqrt(v[1]); dr = std::sqrt(v[2]); dv = std::sqrt(v[3]);
      }
      primal_res = (T)pr; primal_var_norm = (T)pv; dual_res = (T)dr; dual_var_norm = (T)dv;
      T eps_p = eps_primal(), eps_d = eps_dual();
      T rho_prev = rho;
      const T arb_tau = (T)ao.arb_tau, arb_gamma = (T)ao.arb_gamma;
      if ((dual_res < eps_d) && (arb_tau * iteration > arb_l)) { rho *= delta; delta *= arb_gamma; arb_u = (int)iteration; }
      else if ((primal_res < eps_p) && (arb_tau * iteration > arb_u)) { rho /= delta; delta *= arb_gamma; arb_l = (int)iteration; }
      if ((double)std::abs(rho - rho_prev) > 1e-7) {
        const T f = rho_prev / rho;
        for (size_t i = 0; i < n; i++) x_dual[i] = f * x_dual[i];
        for (size_t i = 0; i < m; i++) z_dual[i] = f * z_dual[i];
      }
    }
  }
  // BackendADMM::current_solution (backend_admm.cu:694-743)
  void admm_current_solution(std::vector<T>& px, std::vector<T>& pz, std::vector<T>& dy, std::vector<T>& dw) {
    const size_t m = prob->nrows, n = prob->ncols;
    const T* Sl = prob->left.data(); const T* Tr = prob->right.data();
    for (size_t i = 0; i < n; i++) temp1[i] = -rho * std::pow(Tr[i], (T)-1) * (x_half[i] - x_proj[i] + x_dual[i]);
    dw = temp1;
    for (size_t i = 0; i < m; i++) temp2[i] = -rho * std::pow(Sl[i], (T)1) * (z_half[i] - z_proj[i] + z_dual[i]);
    dy.assign(temp2.begin(), temp2.begin() + m); px = x_half; pz = z_half;
  }

  void perform_iteration() { if (is_admm) admm_iteration(); else pdhg_iteration(); }
  void current_solution(std::vector<T>& px, std::vector<T>& pz, std::vector<T>& dy, std::vector<T>& dw) {
    if (is_admm) admm_current_solution(px, pz, dy, dw); else pdhg_current_solution(px, pz, dy, dw);
  }
  void iterate(int iters) override { have_cur = false; for (int i = 0; i < iters; i++) perform_iteration(); }

  std::vector<T> cur_x, cur_z, cur_y, cur_w;
  bool have_cur = false;

  // Solver::Solve (solver.cu:123-209)
  void solve(int* result, int* iters_done) override {
    int res = 1;   // kStoppedMaxIters
    std::list<double> cb_iters;
    if (num_cback_calls >= 2) {                                  // linspace (common.cu:33-46)
      std::vector<double> ls(num_cback_calls + 1);
      orc_linspace(0, max_iters - 1, num_cback_calls, ls.data());
      cb_iters.assign(ls.begin(), ls.end());
    } else cb_iters.push_back(1e8);
    int i = 0;
    for (i = 0; i < max_iters; i++) {
      perform_iteration();
      T pres = primal_res, dres = dual_res, eps_p = eps_primal(), eps_d = eps_dual();
      bool is_converged = false;
      bool is_stopped = stop_cb ? (stop_cb(cb_user) != 0) : false;
      if ((pres < eps_p) && (dres < eps_d)) is_converged = true;
      if (i >= cb_iters.front() || is_converged || is_stopped || i == (max_iters - 1)) {
        current_solution(cur_x, cur_z, cur_y, cur_w); have_cur = true;
        if (num_cback_calls >= 1) {
          if (verbose)
            std::printf("It %d: Feas_p=%.2e, Eps_p=%.2e, Feas_d=%.2e, Eps_d=%.2e; ", i + 1, (double)pres, (double)eps_p, (double)dres, (double)eps_d);
          if (interm_cb) {
            std::vector<double> dx(cur_x.begin(), cur_x.end()), dy(cur_y.begin(), cur_y.end());
            if (solve_dual) is_converged |= (interm_cb(cb_user, i + 1, dy.data(), dy.size(), dx.data(), dx.size()) != 0);
            else is_converged |= (interm_cb(cb_user, i + 1, dx.data(), dx.size(), dy.data(), dy.size()) != 0);
          } else if (verbose) std::printf("\n");
        }
        cb_iters.pop_front();
      }
      if (is_stopped) { res = 2; i++; break; }
      if (is_converged) { res = 0; i++; break; }
    }
    if (solve_dual) { prob->dualize(); x0.swap(y0); }
    *result = res; *iters_done = i;
  }

  void get(double* ox, double* oz, double* oy, double* ow) override {
    std::vector<T> px, pz, dy, dw;
    if (have_cur) { px = cur_x; pz = cur_z; dy = cur_y; dw = cur_w; }
    else current_solution(px, pz, dy, dw);
    // Solver::cur_*_sol swap under solve_dual (solver.cu:216-246)
    if (solve_dual) { px.swap(dy); pz.swap(dw); }
    if (ox) std::copy(px.begin(), px.end(), ox);
    if (oz) std::copy(pz.begin(), pz.end(), oz);
    if (oy) std::copy(dy.begin(), dy.end(), oy);
    if (ow) std::copy(dw.begin(), dw.end(), ow);
  }
  int last_cg_iters = 0;
  void scalars(double* o) override {
    o[0] = tau; o[1] = sigma; o[2] = theta; o[3] = primal_res; o[4] = dual_res; o[5] = primal_var_norm;
    o[6] = dual_var_norm; o[7] = eps_primal(); o[8] = eps_dual(); o[9] = (double)iteration; o[10] = rho; o[11] = delta; o[12] = last_cg_iters;
  }
};

template <class T> const T* cptr(const void* p) { return static_cast<const T*>(p); }
template <class T> T* mptr(void* p) { return static_cast<T*>(p); }

}  // namespace

struct orc_problem { std::unique_ptr<ProblemBase> p; };
struct orc_prox {
  // description kept in double; instantiated per dtype on demand
  int kind = PK_ELEM; size_t index = 0, size = 0; bool diagsteps = true;
  size_t count = 0, dim = 0; bool interleaved = false; int op = 0, fn = 0;
  std::array<std::vector<double>, 7> coeffs; std::vector<double> a, b, c;
  std::vector<int> perm; std::vector<size_t> inds, inds2; size_t count2 = 0, dim2 = 0; double sum = 0, sum2 = 0, alpha = 1; bool two = false;
  std::unique_ptr<orc_prox> child;
  template <class T> std::shared_ptr<Prox<T>> make() const {
    auto p = std::make_shared<Prox<T>>();
    p->kind = kind; p->index = index; p->size = size; p->diagsteps = diagsteps;
    p->count = count; p->dim = dim; p->interleaved = interleaved; p->op = op; p->fn = fn;
    for (int i = 0; i < 7; i++) p->coeffs[i].assign(coeffs[i].begin(), coeffs[i].end());
    p->a.assign(a.begin(), a.end()); p->b.assign(b.begin(), b.end()); p->c.assign(c.begin(), c.end());
    p->perm = perm; p->inds = inds; p->inds2 = inds2; p->count2 = count2; p->dim2 = dim2; p->sum = (T)sum; p->sum2 = (T)sum2;
    p->alpha = (T)alpha; p->two = two;
    if (child) { auto ch = child->make<T>(); p->child.reset(new Prox<T>(Solver<T>::clone(*ch))); }
    return p;
  }
};
struct orc_solver { std::unique_ptr<SolverBase> s; };

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }
void orc_set_num_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_bind_threads(int on) { return bind_threads(on); }

int orc_grad2d(int dtype, int adjoint, void* res, const void* rhs, size_t nx, size_t ny, size_t L, int lf) {
  ORC_TRY
  if (dtype == 0) grad2d_run<float>(adjoint, mptr<float>(res), cptr<float>(rhs), nx, ny, L, lf);
  else grad2d_run<double>(adjoint, mptr<double>(res), cptr<double>(rhs), nx, ny, L, lf);
  ORC_CATCH
}
int orc_grad3d(int dtype, int adjoint, void* res, const void* rhs, size_t nx, size_t ny, size_t L, int lf) {
  ORC_TRY
  if (dtype == 0) grad3d_run<float>(adjoint, mptr<float>(res), cptr<float>(rhs), nx, ny, L, lf);
  else grad3d_run<double>(adjoint, mptr<double>(res), cptr<double>(rhs), nx, ny, L, lf);
  ORC_CATCH
}
int orc_diags_sort(size_t ndiags, int64_t* ofs, float* fac) {   // block_diags.cu:110-118
  for (size_t i = 0; i < ndiags; i++)
    for (size_t j = i; j < ndiags; j++)
      if (ofs[i] > ofs[j]) { std::swap(ofs[i], ofs[j]); std::swap(fac[i], fac[j]); }
  return 0;
}
int orc_diags(int dtype, int adjoint, void* res, const void* rhs, size_t nrows, size_t ncols, size_t ndiags,
              const int64_t* ofs, const float* fac, int quirk) {
  ORC_TRY
  if (dtype == 0) diags_run<float>(adjoint, mptr<float>(res), cptr<float>(rhs), nrows, ncols, ndiags, ofs, fac, quirk);
  else diags_run<double>(adjoint, mptr<double>(res), cptr<double>(rhs), nrows, ncols, ndiags, ofs, fac, quirk);
  ORC_CATCH
}
int orc_csr2csc(int dtype, int n, int m, int nz, const void* a, const int32_t* ci, const int32_t* rs,
                void* csc_a, int32_t* ri, int32_t* cs) {
  ORC_TRY
  if (dtype == 0) csr2csc_run<float>(n, m, nz, cptr<float>(a), ci, rs, mptr<float>(csc_a), ri, cs);
  else csr2csc_run<double>(n, m, nz, cptr<double>(a), ci, rs, mptr<double>(csc_a), ri, cs);
  ORC_CATCH
}
int orc_csr_spmv_acc(int dtype, void* res, const void* rhs, int nrows, const void* val, const int32_t* ptr, const int32_t* ind) {
  ORC_TRY
  if (dtype == 0) csr_spmv_acc<float>(mptr<float>(res), cptr<float>(rhs), nrows, cptr<float>(val), ptr, ind);
  else csr_spmv_acc<double>(mptr<double>(res), cptr<double>(rhs), nrows, cptr<double>(val), ptr, ind);
  ORC_CATCH
}
int orc_prox_elem(int dtype, int op, int fn, void* res, const void* arg, const void* tau_diag, double tau, int invert,
                  size_t count, size_t dim, int interleaved, const void* const* coeff_ptr, const double* coeff_val) {
  ORC_TRY
  if (dtype == 0) {
    const float* cp[7]; float cv[7];
    for (int i = 0; i < 7; i++) { cp[i] = cptr<float>(coeff_ptr ? coeff_ptr[i] : nullptr); cv[i] = (float)coeff_val[i]; }
    prox_elem_run<float>(op, fn, mptr<float>(res), cptr<float>(arg), cptr<float>(tau_diag), (float)tau, invert, count, dim, interleaved, cp, cv);
  } else {
    const double* cp[7]; double cv[7];
    for (int i = 0; i < 7; i++) { cp[i] = cptr<double>(coeff_ptr ? coeff_ptr[i] : nullptr); cv[i] = coeff_val[i]; }
    prox_elem_run<double>(op, fn, mptr<double>(res), cptr<double>(arg), cptr<double>(tau_diag), tau, invert, count, dim, interleaved, cp, cv);
  }
  ORC_CATCH
}
int orc_prox_epi_quad(int dtype, void* res, const void* arg, size_t count, size_t dim, const void* a_ptr, double a_val,
                      const void* b_ptr, const void* c_ptr, double c_val) {
  ORC_TRY
  if (dtype == 0) epi_quad_run<float>(mptr<float>(res), cptr<float>(arg), count, dim, cptr<float>(a_ptr), (float)a_val, cptr<float>(b_ptr), cptr<float>(c_ptr), (float)c_val);
  else epi_quad_run<double>(mptr<double>(res), cptr<double>(arg), count, dim, cptr<double>(a_ptr), a_val, cptr<double>(b_ptr), cptr<double>(c_ptr), c_val);
  ORC_CATCH
}
void orc_glibc_rand_fill(unsigned seed, size_t n, int32_t* out) {
  GlibcRand g(seed);
  for (size_t i = 0; i < n; i++) out[i] = g.next();
}
int orc_linspace(double start, double end, int num_in, double* out) {   // common.cu:33-46
  double num = (double)num_in;
  double delta = (end - start) / (num - 1);
  int k = 0;
  for (int i = 0; i < num; ++i) out[k++] = start + delta * i;
  out[k++] = end;
  return 0;
}

// ---- problem ----
orc_problem* orc_problem_create(int dtype, size_t nrows, size_t ncols) {
  orc_problem* h = new orc_problem;
  if (dtype == 0) h->p.reset(new Problem<float>()); else h->p.reset(new Problem<double>());
  h->p->dtype = dtype; h->p->nrows = nrows; h->p->ncols = ncols;
  return h;
}
void orc_problem_destroy(orc_problem* h) { delete h; }

#define WITH_PROB(h, ...)                                                          \
  if ((h)->p->dtype == 0) { typedef float T; Problem<T>& P = static_cast<Problem<T>&>(*(h)->p); (void)P; __VA_ARGS__ } \
  else { typedef double T; Problem<T>& P = static_cast<Problem<T>&>(*(h)->p); (void)P; __VA_ARGS__ }

int orc_problem_add_block_grad(orc_problem* h, int is3d, size_t row, size_t col, size_t nx, size_t ny, size_t L, int lf) {
  ORC_TRY
  WITH_PROB(h, Block<T> b; b.kind = is3d ? BK_GRAD3D : BK_GRAD2D; b.row = row; b.col = col;
            b.nrows = nx * ny * L * (is3d ? 3 : 2); b.ncols = nx * ny * L; b.nx = nx; b.ny = ny; b.L = L; b.lf = lf;
            P.blocks.push_back(b);)
  ORC_CATCH
}
int orc_problem_add_block_diags(orc_problem* h, size_t row, size_t col, size_t nrows, size_t ncols, size_t ndiags,
                                const int64_t* offsets, const double* factors) {
  ORC_TRY
  WITH_PROB(h, Block<T> b; b.kind = BK_DIAGS; b.row = row; b.col = col; b.nrows = nrows; b.ncols = ncols; b.ndiags = ndiags;
            b.ofs.assign(offsets, offsets + ndiags);
            std::vector<T> ft(factors, factors + ndiags);             // factory.cpp:580 (double -> real)
            b.fac.assign(ft.begin(), ft.end());                       // block_diags.cu:108 (real -> float)
            orc_diags_sort(ndiags, b.ofs.data(), b.fac.data());
            P.blocks.push_back(b);)
  ORC_CATCH
}
int orc_problem_add_block_sparse_csc(orc_problem* h, size_t row, size_t col, int nrows, int ncols, int nnz,
                                     const double* val, const int32_t* jc, const int32_t* ir) {
  ORC_TRY
  WITH_PROB(h, Block<T> b; b.kind = BK_SPARSE; b.row = row; b.col = col; b.nrows = nrows; b.ncols = ncols; b.nnz = nnz;
            // BlockSparse::CreateFromCSC (block_sparse.cu:34-68): the CSC arrays ARE the CSR of K^T
            b.val_t.assign(val, val + nnz); b.ptr_t.assign(jc, jc + ncols + 1); b.ind_t.assign(ir, ir + nnz);
            b.val.resize(nnz); b.ind.resize(nnz); b.ptr.resize(nrows + 1);
            csr2csc_run<T>(ncols, nrows, nnz, b.val_t.data(), b.ind_t.data(), b.ptr_t.data(), b.val.data(), b.ind.data(), b.ptr.data());
            P.blocks.push_back(b);)
  ORC_CATCH
}
int orc_problem_add_block_kron_csc(orc_problem* h, int id_first, size_t row, size_t col, size_t diaglength, int nrows, int ncols, int nnz,
                                   const double* val, const int32_t* jc, const int32_t* ir) {
  ORC_TRY
  WITH_PROB(h, Block<T> b; b.kind = id_first ? BK_ID_KRON_SPARSE : BK_SPARSE_KRON_ID; b.row = row; b.col = col;
            b.nrows = (size_t)nrows * diaglength; b.ncols = (size_t)ncols * diaglength; b.nnz = nnz;
            b.diaglength = diaglength; b.mat_nrows = nrows; b.mat_ncols = ncols;
            // CreateFromCSC (block_sparse_kron_id.cu:59-99): double -> real (factory) -> float (:79)
            std::vector<T> vt(val, val + nnz);
            b.fval_t.assign(vt.begin(), vt.end()); b.ptr_t.assign(jc, jc + ncols + 1); b.ind_t.assign(ir, ir + nnz);
            b.fval.resize(nnz); b.ind.resize(nnz); b.ptr.resize(nrows + 1);
            csr2csc_run<float>(ncols, nrows, nnz, b.fval_t.data(), b.ind_t.data(), b.ptr_t.data(), b.fval.data(), b.ind.data(), b.ptr.data());
            P.blocks.push_back(b);)
  ORC_CATCH
}
int orc_problem_add_block_zero(orc_problem* h, size_t row, size_t col, size_t nrows, size_t ncols) {
  ORC_TRY
  WITH_PROB(h, Block<T> b; b.kind = BK_ZERO; b.row = row; b.col = col; b.nrows = nrows; b.ncols = ncols; P.blocks.push_back(b);)
  ORC_CATCH
}

orc_prox* orc_prox_elem_create(int op, int fn, size_t idx, size_t count, size_t dim, int interleaved, int diagsteps,
                               const double* const* coeff, const size_t* coeff_len) {
  orc_prox* p = new orc_prox;
  p->kind = PK_ELEM; p->op = op; p->fn = fn; p->index = idx; p->count = count;
  p->dim = (op == ORC_OP_1D) ? 1 : dim;      // prox_elem_operation.hpp:83 (kDim<=0 ? dim : kDim)
  p->size = p->count * p->dim; p->interleaved = interleaved; p->diagsteps = diagsteps;
  for (int i = 0; i < 7; i++) p->coeffs[i].assign(coeff[i], coeff[i] + coeff_len[i]);
  return p;
}
orc_prox* orc_prox_moreau_create(orc_prox* child) {
  orc_prox* p = new orc_prox;
  p->kind = PK_MOREAU; p->index = child->index; p->size = child->size; p->diagsteps = child->diagsteps;
  p->child.reset(child);
  return p;
}
orc_prox* orc_prox_zero_create(size_t idx, size_t size) {
  orc_prox* p = new orc_prox; p->kind = PK_ZERO; p->index = idx; p->size = size; p->diagsteps = true; return p;
}
orc_prox* orc_prox_epi_quad_create(size_t idx, size_t count, size_t dim, int interleaved, int diagsteps,
                                   const double* a, size_t na, const double* b, size_t nb, const double* c, size_t nc) {
  orc_prox* p = new orc_prox;
  p->kind = PK_EPI_QUAD; p->index = idx; p->count = count; p->dim = dim; p->size = count * dim;
  p->interleaved = interleaved; p->diagsteps = diagsteps;
  p->a.assign(a, a + na); p->b.assign(b, b + nb); p->c.assign(c, c + nc);
  return p;
}
orc_prox* orc_prox_elem_nocoeff_create(int op, size_t idx, size_t count, size_t dim, int interleaved, int diagsteps) {
  if (op != ORC_OP_IND_SUM && op != ORC_OP_IND_SIMPLEX) return nullptr;
  orc_prox* p = new orc_prox;
  p->kind = PK_ELEM; p->op = op; p->index = idx; p->count = count; p->dim = dim; p->size = count * dim;
  p->interleaved = interleaved; p->diagsteps = diagsteps;
  for (int i = 0; i < 7; i++) p->coeffs[i].assign(1, 0.0);
  return p;
}
// ProxTransform copies index/size/diagsteps of the inner prox (prox_transform.cu:107: Prox<T>(*inner_fn))
orc_prox* orc_prox_transform_create(orc_prox* child, const double* const* coeff, const size_t* coeff_len) {
  orc_prox* p = new orc_prox;
  p->kind = PK_TRANSFORM; p->index = child->index; p->size = child->size; p->diagsteps = child->diagsteps;
  for (int i = 0; i < 5; i++) p->coeffs[i].assign(coeff[i], coeff[i] + coeff_len[i]);
  for (int i = 5; i < 7; i++) p->coeffs[i].assign(1, 0.0);
  p->child.reset(child);
  return p;
}
orc_prox* orc_prox_permute_create(orc_prox* child, const int* perm, size_t n) {
  orc_prox* p = new orc_prox;
  p->kind = PK_PERMUTE; p->index = child->index; p->size = child->size; p->diagsteps = child->diagsteps;
  p->perm.assign(perm, perm + n);
  p->child.reset(child);
  return p;
}
orc_prox* orc_prox_halfspace_create(size_t idx, size_t count, size_t dim, int interleaved, int diagsteps, const double* a, size_t na,
                                    const double* b, size_t nb) {
  orc_prox* p = new orc_prox;
  p->kind = PK_HALFSPACE; p->index = idx; p->count = count; p->dim = dim; p->size = count * dim;
  p->interleaved = interleaved; p->diagsteps = diagsteps;
  p->a.assign(a, a + na); p->b.assign(b, b + nb);
  return p;
}
orc_prox* orc_prox_soc_create(size_t idx, size_t count, size_t dim, int interleaved, int diagsteps, double alpha) {
  orc_prox* p = new orc_prox;
  p->kind = PK_SOC; p->index = idx; p->count = count; p->dim = dim; p->size = count * dim;
  p->interleaved = interleaved; p->diagsteps = diagsteps; p->alpha = alpha;
  return p;
}
// ProxIndSum: Prox<T>(index, size, true) (prox_ind_sum.hpp:41); count = inds / dim (factory.cpp:466)
orc_prox* orc_prox_ind_sum_create(size_t idx, size_t size, size_t dim, const size_t* inds, size_t ninds, double sum,
                                  size_t dim2, const size_t* inds2, size_t ninds2, double sum2) {
  orc_prox* p = new orc_prox;
  p->kind = PK_IND_SUM; p->index = idx; p->size = size; p->diagsteps = true;
  p->dim = dim; p->inds.assign(inds, inds + ninds); p->count = dim ? ninds / dim : 0; p->sum = sum;
  if (inds2) { p->two = true; p->dim2 = dim2; p->inds2.assign(inds2, inds2 + ninds2); p->count2 = dim2 ? ninds2 / dim2 : 0; p->sum2 = sum2; }
  return p;
}
void orc_prox_destroy(orc_prox* p) { delete p; }
size_t orc_prox_size(const orc_prox* p) { return p->size; }
int orc_prox_eval(orc_prox* p, int dtype, void* res, const void* arg, const void* tau_diag, double tau) {
  ORC_TRY
  // eval_prox.m calls prox(0, size(arg,1)); Prox::Eval offsets by index (prox.cu:27-43)
  if (dtype == 0) { auto q = p->make<float>(); q->initialize(); q->eval(mptr<float>(res), cptr<float>(arg), cptr<float>(tau_diag), (float)tau); }
  else { auto q = p->make<double>(); q->initialize(); q->eval(mptr<double>(res), cptr<double>(arg), cptr<double>(tau_diag), tau); }
  ORC_CATCH
}
int orc_problem_add_prox(orc_problem* h, int which, orc_prox* p) {
  ORC_TRY
  std::unique_ptr<orc_prox> own(p);
  WITH_PROB(h, auto q = p->make<T>();
            switch (which) { case ORC_PROX_G: P.prox_g.push_back(q); break; case ORC_PROX_F: P.prox_f.push_back(q); break;
                             case ORC_PROX_GSTAR: P.prox_gstar.push_back(q); break; case ORC_PROX_FSTAR: P.prox_fstar.push_back(q); break;
                             default: throw OrcError("bad prox list id"); })
  ORC_CATCH
}
int orc_problem_set_scaling_alpha(orc_problem* h, double alpha) { WITH_PROB(h, P.scaling_type = 0; P.scaling_alpha = (T)alpha;) return 0; }
int orc_problem_set_scaling_identity(orc_problem* h) { WITH_PROB(h, P.scaling_type = 1;) return 0; }
int orc_problem_set_scaling_custom(orc_problem* h, const double* l, size_t nl, const double* r, size_t nr) {
  // SetScalingCustom squares the user vectors (problem.cu:357-380)
  WITH_PROB(h, P.scaling_type = 2; P.left.resize(nl); P.right.resize(nr);
            for (size_t i = 0; i < nl; i++) { T v = (T)l[i]; P.left[i] = v * v; }
            for (size_t i = 0; i < nr; i++) { T v = (T)r[i]; P.right[i] = v * v; })
  return 0;
}
int orc_problem_initialize(orc_problem* h) { ORC_TRY WITH_PROB(h, P.initialize();) ORC_CATCH }
int orc_problem_get_scaling(const orc_problem* h, double* l, double* r) {
  WITH_PROB(h, std::copy(P.left.begin(), P.left.end(), l); std::copy(P.right.begin(), P.right.end(), r);)
  return 0;
}
int orc_problem_normest(orc_problem* h, double tol, int max_iters, double* out) {
  ORC_TRY WITH_PROB(h, *out = (double)P.normest((T)tol, max_iters);) ORC_CATCH
}
size_t orc_problem_nrows(const orc_problem* h) { return h->p->nrows; }
size_t orc_problem_ncols(const orc_problem* h) { return h->p->ncols; }
int orc_linop_eval(orc_problem* h, int adjoint, void* res, const void* rhs) {
  ORC_TRY
  WITH_PROB(h, P.linop_initialize();
            if (!adjoint) P.K_eval(mptr<T>(res), P.lin_nrows, cptr<T>(rhs), 0, false);
            else P.K_eval(mptr<T>(res), P.lin_ncols, cptr<T>(rhs), 0, true);)
  ORC_CATCH
}
int orc_linop_sums(orc_problem* h, double alpha, double* rowsum, double* colsum) {
  ORC_TRY
  WITH_PROB(h, P.linop_initialize();
            for (size_t r = 0; r < P.lin_nrows; r++) rowsum[r] = (double)P.K_row_sum(r, (T)alpha);
            for (size_t c = 0; c < P.lin_ncols; c++) colsum[c] = (double)P.K_col_sum(c, (T)alpha);)
  ORC_CATCH
}

// ---- solver ----
}  // extern "C"
template <class T>
static Solver<T>* make_solver(Problem<T>* P, const orc_solver_opts* so) {
  Solver<T>* s = new Solver<T>();
  s->prob = P;
  s->tol_rel_primal = (T)so->tol_rel_primal; s->tol_rel_dual = (T)so->tol_rel_dual;
  s->tol_abs_primal = (T)so->tol_abs_primal; s->tol_abs_dual = (T)so->tol_abs_dual;
  s->max_iters = so->max_iters; s->num_cback_calls = so->num_cback_calls;
  s->verbose = so->verbose; s->solve_dual = so->solve_dual;
  if (so->x0 && so->nx0) s->x0.assign(so->x0, so->x0 + so->nx0);
  if (so->y0 && so->ny0) s->y0.assign(so->y0, so->y0 + so->ny0);
  return s;
}
extern "C" {
orc_solver* orc_solver_create_pdhg(orc_problem* h, const orc_pdhg_opts* po, const orc_solver_opts* so) {
  orc_solver* r = new orc_solver;
  WITH_PROB(h, Solver<T>* s = make_solver<T>(&P, so); s->is_admm = false; s->po = *po; r->s.reset(s);)
  return r;
}
orc_solver* orc_solver_create_admm(orc_problem* h, const orc_admm_opts* ao, const orc_solver_opts* so) {
  orc_solver* r = new orc_solver;
  WITH_PROB(h, Solver<T>* s = make_solver<T>(&P, so); s->is_admm = true; s->ao = *ao; r->s.reset(s);)
  return r;
}
void orc_solver_destroy(orc_solver* s) { delete s; }
int orc_solver_set_callbacks(orc_solver* s, orc_interm_cb icb, orc_stop_cb scb, void* user) {
  s->s->interm_cb = icb; s->s->stop_cb = scb; s->s->cb_user = user; return 0;
}
int orc_solver_set_allreduce(orc_solver* s, orc_allreduce_cb cb, void* user, size_t gr, size_t gc) {
  s->s->allreduce = cb; s->s->ar_user = user; s->s->g_nrows = gr; s->s->g_ncols = gc; return 0;
}
int orc_solver_initialize(orc_solver* s) { ORC_TRY s->s->initialize(); ORC_CATCH }
int orc_solver_iterate(orc_solver* s, int iters) { ORC_TRY s->s->iterate(iters); ORC_CATCH }
int orc_solver_rehome(orc_solver* s) { ORC_TRY s->s->rehome(); ORC_CATCH }
int orc_solver_solve(orc_solver* s, int* result, int* iters_done) { ORC_TRY s->s->solve(result, iters_done); ORC_CATCH }
int orc_solver_get(orc_solver* s, double* x, double* z, double* y, double* w) { ORC_TRY s->s->get(x, z, y, w); ORC_CATCH }
int orc_solver_scalars(orc_solver* s, double* out) { ORC_TRY s->s->scalars(out); ORC_CATCH }

}  // extern "C"
