// device_math.hpp -- scalar proximal arithmetic shared by the generic and the fused kernels.
//
// Written for gfx950 from the formulas of the reference (cited per function); the fp32 build keeps
// the reference's implicit promotions to double wherever a double literal appears in its
// expressions, so results round exactly as the reference's do.  Divisions by a wave-uniform
// operand that equals 1 are elided (x/1 == x exactly): with scalar coefficients e = 0 and a = 1
// (the ROF case) this removes every fp64 division but one from the hot loops.
#pragma once
#include <hip/hip_runtime.h>

#include "prost/prox/elemop/function_1d.hpp"
#include "prost_hip.h"

namespace prost_hip {

// t_abs / t_sqrt / t_pow ..., div1 and the scalar maps f1d_* live in the public header (plugin authors compose them too)
using namespace prost::elemop;

// ---- fp64: the compiler's own IEEE expansions of `/` and sqrt without their range handling -----------------
// hipcc expands a / b into v_div_scale x 2, v_rcp_f64, two Newton steps on the reciprocal (4 fma), the quotient estimate, its
// residual, v_div_fmas, v_div_fixup: 11 fp64-rate instructions PER QUOTIENT, and does not share the reciprocal between quotients
// with the same divisor; sqrt is v_rsq_f64 + 2 products + 7 fma inside a range scaling (compare, 2 x ldexp, class test, 4 selects).
// The forms below are those sequences minus v_div_scale / v_div_fmas' scaling / v_div_fixup resp. the range scaling: the SAME bits
// wherever the dropped steps are the identity -- operands and results well inside the exponent range, checked by the callers
// (f64_mid: |v| in [2^-500, 2^500]; quotients of two such values are normal) -- i.e. the correctly rounded quotient / root
// (kernels_selftest.hip compares them with `/` and sqrt on the device).  The reciprocal of a divisor is formed ONCE:
// 5 instructions, then 3 per quotient; a wave-uniform divisor pays the 5 once per launch.
__device__ __forceinline__ bool f64_mid(double v) {
  const unsigned e = ((unsigned)((unsigned long long)__double_as_longlong(v) >> 52)) & 0x7FFu;
  return e - (1023u - 500u) <= 1000u;
}
__device__ __forceinline__ double rcp_newton2(double d) {           // the reciprocal as a / d's expansion refines it
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  return y;
}
__device__ __forceinline__ double div_mid(double n, double d, double y) {   // n / d for f64_mid n (or 0), d; y = rcp_newton2(d)
  const double q0 = n * y;
  const double r = __builtin_fma(-d, q0, n);
  return __builtin_fma(r, y, q0);
}
// ---- exact (float)((double)x / D) for a wave-uniform divisor D ------------------------------
// The reference's fp32 build divides in double wherever a double literal appears (e.g.
// Function1DSquare: x0 / (1. + tau)).  A hardware fp64 division is ~35 VALU instructions; with a
// uniform divisor the reciprocal rD = 1/D is computed once per wave and the quotient costs
// 2 fma + 1 mul.  q1 below is within 1 ulp(double) of RN_double(x/D); rounding q1 to float gives
// the reference's float unless q1 lies within 2 ulp(double) of a float rounding boundary -- that
// case (probability ~2^-27 per element) takes the full division, so the result is ALWAYS
// identical to (float)((double)x / D).
struct UniformDiv {
  double D, rD;
};
__device__ __forceinline__ UniformDiv make_uniform_div(double D) { UniformDiv u; u.D = D; u.rD = 1.0 / D; return u; }
__device__ __forceinline__ float div_to_float_exact(float x, const UniformDiv& u) {
  if (u.D == 1.0) return x;
  const double xd = (double)x;
  const double q0 = xd * u.rD;
  const double rem = __builtin_fma(-q0, u.D, xd);
  const double q1 = __builtin_fma(rem, u.rD, q0);
  const unsigned long long bits = (unsigned long long)__double_as_longlong(q1);
  const int low = (int)(bits & 0x1FFFFFFFull);                 // the 29 mantissa bits a float drops
  const int bexp = (int)((bits >> 52) & 0x7FF);                  // biased double exponent
  const bool near_tie = (low >= (1 << 28) - 2) && (low <= (1 << 28) + 2);
  const bool normal_float = bexp > 1023 - 126 && bexp < 1023 + 127;
  if (__builtin_expect((near_tie || !normal_float) && q1 != 0.0, 0)) return (float)(xd / u.D);
  return (float)q1;
}
__device__ __forceinline__ double div_to_float_exact(double x, const UniformDiv& u) {
  if (u.D == 1.0) return x;
  const double y = rcp_newton2(u.D);                              // (wave-uniform: hoisted out of the callers' loops)
  return (f64_mid(u.D) && (f64_mid(x) || x == 0.0)) ? div_mid(x, u.D, y) : x / u.D;
}
// The same for VEC values with straight-line code, the guards evaluated as WAVE MASKS (the compares write SGPR pairs,
// the OR runs on the scalar unit: no v_cndmask / v_or per element) and without the residual correction step: q0 = RN(x * rD) with rD = RN(1/D) is within
// 2^-52 |q| < 2 ulp(double) of x / D, the reference's RN_double(x / D) within 1/2 ulp, so the two doubles differ by at
// most 2 ulp and round to the same float unless a float rounding boundary lies within 4 ulp of q0 (window below:
// +-4 ulp, probability 2^-26 per element) -- those elements, and results that are not normal floats, take the division.
// The fallback is taken by the whole wave (the mask is wave-uniform); recomputing an element whose fast result was
// already right gives the same bits.
template <int VEC>
__device__ __forceinline__ void div_to_float_exact_vec(const float (&x)[VEC], const UniformDiv& u, float (&out)[VEC]) {
  unsigned long long odd = 0;
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    const double q0 = (double)x[j] * u.rD;
    out[j] = (float)q0;
    const unsigned low = (unsigned)((unsigned long long)__double_as_longlong(q0)) & 0x1FFFFFFFu;
    odd |= __builtin_amdgcn_ballot_w64((low - ((1u << 28) - 4u)) <= 8u);
    odd |= __builtin_amdgcn_ballot_w64(!__builtin_isnormal(out[j]));
  }
  if (__builtin_expect(odd != 0, 0)) {
    // second look (rare; always taken by waves with padding lanes, whose x is 0): a zero result of a zero x is exact
    unsigned long long slow = 0;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const double q0 = (double)x[j] * u.rD;
      const unsigned low = (unsigned)((unsigned long long)__double_as_longlong(q0)) & 0x1FFFFFFFu;
      slow |= __builtin_amdgcn_ballot_w64((low - ((1u << 28) - 4u)) <= 8u || (!__builtin_isnormal(out[j]) && x[j] != 0.0f));
    }
    if (slow != 0) {
#pragma unroll
      for (int j = 0; j < VEC; j++) out[j] = (float)((double)x[j] / u.D);
    }
  }
}
// fp64: the short quotient for every element, the range guard as a wave mask, the full division for the whole wave if any lane needs it
template <int VEC>
__device__ __forceinline__ void div_to_float_exact_vec(const double (&x)[VEC], const UniformDiv& u, double (&out)[VEC]) {
  const double y = rcp_newton2(u.D);
  unsigned long long odd = __builtin_amdgcn_ballot_w64(!f64_mid(u.D));
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    out[j] = div_mid(x[j], u.D, y);
    odd |= __builtin_amdgcn_ballot_w64(!(f64_mid(x[j]) || x[j] == 0.0));
  }
  if (__builtin_expect(odd != 0, 0)) {
#pragma unroll
    for (int j = 0; j < VEC; j++) out[j] = x[j] / u.D;
  }
}

// ---- correctly rounded fp32 division / square root with fewer instructions ---------------------
// A VALU instruction costs one issue slot of 4 cycles whatever it does, and the compiler's IEEE
// expansions are long: x / y = 11 instructions (v_div_scale x2, v_rcp, 6 fma/mul, v_div_fmas,
// v_div_fixup), sqrtf = 16 (range scaling + refinement + class fix-up).  The two-iteration kernel
// is bound by exactly that, so its hot path uses the forms below.  Both return the correctly rounded
// result, i.e. the same bits as `/` and sqrtf (IEEE results are unique).
//
// Division by a shared denominator through ONE double reciprocal: r = 1/(double)d refined to
// ~2^-52, q = (float)((double)n * r).  |n r - n/d| < 2^-51 |n/d|, while the exact quotient of two
// 24-bit floats is never within 2^-49 (relative) of a float rounding boundary (n = m d with m a
// 25-bit midpoint is impossible, and n - m d is a non-zero multiple of the product's last place),
// so rounding n r gives RN(n/d) -- including subnormal quotients and signed zeros; d = +-0, +-inf
// or NaN are NOT supported (callers substitute / fall back).
__device__ __forceinline__ double rcp_refined(float d) {
  const double dd = (double)d;
  double r = __builtin_amdgcn_rcp(dd);                     // v_rcp_f64: 2^-23 relative
  double e = __builtin_fma(-dd, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-dd, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
__device__ __forceinline__ double rcp_refined(double d) { return 1.0 / d; }
// the same from the single-precision reciprocal estimate (v_rcp_f32, 1 ulp, converted) instead of v_rcp_f64: two Newton
// steps take 2^-22.4 to below 2^-52 all the same
__device__ __forceinline__ double rcp_refined_s(float d) {
  const double dd = (double)d;
  double r = (double)__builtin_amdgcn_rcpf(d);
  double e = __builtin_fma(-dd, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-dd, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
__device__ __forceinline__ double rcp_refined_s(double d) { return 1.0 / d; }
__device__ __forceinline__ float mul_rcp(float n, double r) { return (float)((double)n * r); }
// mul_rcp(n, r) with the quotient of a zero numerator forced to +0, in one instruction less than "+ 0.0f" afterwards:
// fma(n, r, +0.0) rounds exactly like the product and turns a -0 product (n = +-0) into +0 already in double.  (A
// non-zero quotient that underflows to zero keeps its sign, like n / d itself.)
__device__ __forceinline__ float mul_rcp_pz(float n, double r) { return (float)__builtin_fma((double)n, r, 0.0); }
__device__ __forceinline__ double mul_rcp_pz(double n, double r) { return n * r + 0.0; }
// min(t, 0) for t that is not NaN (callers: the range-checked fast paths): == (t > 0 ? 0 : t) including t = +-0
__device__ __forceinline__ float min0(float t) { return __builtin_fminf(t, 0.0f); }
__device__ __forceinline__ double min0(double t) { return t > 0.0 ? 0.0 : t; }
// ---- tolerance-class arithmetic (prost_hip_fused_desc.arith = PROST_HIP_ARITH_FMAD) -----------------------
// one instruction each: v_min_f32 / v_max_f32 (operands are never NaN where these are used) and v_rsq_f32 (1 ulp; +inf for +0,
// which the callers' min(b * rsq, 1) turns into the factor 1 on a zero vector)
__device__ __forceinline__ float t_min(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float t_max(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double t_min(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double t_max(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float t_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double t_rsq(double x) { return 1.0 / __builtin_sqrt(x); }
// value of the neighbouring lane of the 64-lane wavefront: ONE v_mov_b32 with a DPP wavefront shift instead of a
// ds_bpermute (address VALU + LDS round trip).  lane_up: lane i reads lane i - 1 (lane 0 gets 0); lane_down: lane i
// reads lane i + 1 (lane 63 gets 0).
__device__ __forceinline__ float lane_up(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false)); }
__device__ __forceinline__ float lane_down(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, false)); }
__device__ __forceinline__ double lane_up(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xF, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lane_down(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xF, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// n / d for many n and one (typically wave-uniform) d: exact float quotient via the double reciprocal,
// plain division for double
template <class T> struct SharedDivisor;
template <> struct SharedDivisor<float> {
  double r;
  __device__ __forceinline__ explicit SharedDivisor(float d) : r(rcp_refined(d)) {}
  __device__ __forceinline__ float div(float n) const { return mul_rcp(n, r); }
};
template <> struct SharedDivisor<double> {
  double d, y;
  bool mid;
  __device__ __forceinline__ explicit SharedDivisor(double d_) : d(d_), y(rcp_newton2(d_)), mid(f64_mid(d_)) {}
  __device__ __forceinline__ double div(double n) const { return (mid && (f64_mid(n) || n == 0.0)) ? div_mid(n, d, y) : n / d; }
};
// ---- the same quotient in single precision only -------------------------------------------------
// fp64 instructions and the f32 <-> f64 conversions issue at half the fp32 rate on this part (16 vs 32 lanes per
// cycle and SIMD), v_rcp_f64 far below that; the double-reciprocal form above spends ~70 of its issue cycles there.
// This is the compiler's own IEEE expansion of `/` (v_rcp_f32, one Newton step on the reciprocal, quotient estimate,
// two residual corrections) WITHOUT its v_div_scale / v_div_fmas / v_div_fixup range handling: identical bits whenever
// that handling would be the identity -- d and 1/d normal, the quotient not subnormal, exponent(n) - exponent(d) < 96,
// |n| >= 2^-103 -- and for n = +-0 (result +0: the callers want the sign of a zero quotient dropped).  Callers keep
// d in [2^-48, 2^63] and check the QUOTIENT: |q| >= 2^-55 implies |n| >= 2^-103 and a normal quotient.
struct RcpF32 { float d, y; };
__device__ __forceinline__ RcpF32 rcp_f32_newton(float d) {
  RcpF32 r;
  r.d = d;
  const float y0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, y0, 1.0f);
  r.y = __builtin_fmaf(e, y0, y0);
  return r;
}
__device__ __forceinline__ float div_f32_unscaled(float n, const RcpF32& r) {
  const float q0 = n * r.y;
  const float r0 = __builtin_fmaf(-r.d, q0, n);
  const float q1 = __builtin_fmaf(r0, r.y, q0);
  const float r1 = __builtin_fmaf(-r.d, q1, n);
  return __builtin_fmaf(r1, r.y, q1);
}
constexpr float kDivF32MinQuotient = 2.7755575615628914e-17f;      // 2^-55

// sqrtf(x) for x in [2^-96, 2^126]: v_sqrt_f32 (1 ulp) + the compiler's own +-1 ulp residual test,
// without the 2^32 range scaling and the zero / infinity fix-up that the general expansion carries
__device__ __forceinline__ float sqrt_midrange(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
  const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
  s = r_dn <= 0.0f ? s_dn : s;
  s = r_up > 0.0f ? s_up : s;
  return s;
}
// sqrt(x) for x in [2^-500, 2^500] (f64_mid): the compiler's expansion without its range scaling and its zero / infinity select
__device__ __forceinline__ double sqrt_midrange(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return g;
}

// ---- the dual step's prox of the ROF / TV shapes, straight-line --------------------------------
// ElemOperationNorm2<Function1DIndLeq0> with scalar a = 1, d = 0, e = 0 (host-checked; elem_operation_norm2.hpp:40-88,
// function_1d.hpp:75-87) on the NC gradient components of VEC pixels:
//     out_i = pr v_i / ||v||,  pr = min(||v|| - b, 0) + b;   ||v|| = 0 -> 0
// nv = the squared norms as the reference accumulates them, av = the components.  Norms up to 2^126 take the short
// correctly rounded forms above (sqrt without range scaling, ONE refined double reciprocal per pixel, the products
// rounded through fma(n, r, +0) so that zero quotients come out as +0).  A zero norm runs the same code with
// ||v|| := 2^-48: then pr = RN(RN(2^-48 - b) + b) = 0 for every radius b >= 2^-20, the numerators are +-0 and the
// results +0, the value the reference writes.  The same holds for non-zero norms below 2^-96 (squares of
// subnormal-range components): the reference computes pr = 0 and quotients +-0 there as well, so with
// `tiny_is_zero` (= b >= 2^-20, wave-uniform) they need no separate path; results are equal in value to the
// reference's, zeros may differ in sign.  Without it such norms, like norms above 2^126 and NaNs, take the general
// expansions.
template <class T, int NC, int VEC>
__device__ __forceinline__ void norm2_leq0_fast(const T (&nv)[VEC], const T (&av)[NC][VEC], T b, bool tiny_is_zero, T (&out)[NC][VEC]) {
  constexpr float kLo = 1.2621774483536189e-29f;             // 2^-96
  T nmax = 0;
#pragma unroll
  for (int j = 0; j < VEC; j++) nmax = nv[j] > nmax ? nv[j] : nmax;
  bool mid = sizeof(T) == 4 && nmax <= (T)8.507059173023462e37f;
  if (!tiny_is_zero) {
    unsigned tmin = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < VEC; j++) tmin = min(tmin, (unsigned)__float_as_int((float)nv[j]) - 1u);     // 0 -> 0xFFFFFFFF: a zero norm is fine
    mid = mid && tmin >= (unsigned)__float_as_int(kLo) - 1u;
  }
  if (__builtin_expect(mid, 1)) {
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const T nrm = sqrt_midrange(nv[j] > (T)kLo ? nv[j] : (T)kLo);
      const T pr = min0(nrm - b) + b;
      const auto r = rcp_refined(nrm);
#pragma unroll
      for (int i = 0; i < NC; i++) out[i][j] = mul_rcp_pz(pr * av[i][j], r);
    }
  } else {
    if constexpr (sizeof(T) == 8) {
      // fp64: the short square root, ONE refined reciprocal per pixel and three instructions per quotient (f64_mid forms above) --
      // same bits as the expansions below wherever the squared norm and the numerators are zero or well inside the exponent
      // range; a wave in which any lane is not (the guard is a wave mask) takes the expansions
      unsigned long long odd = 0;
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const bool nz = nv[j] > 0;
        const double x = nz ? (double)nv[j] : 1.0;
        odd |= __builtin_amdgcn_ballot_w64(!f64_mid(x));
        const double nrm = sqrt_midrange(x);
        const double t = nrm - (double)b;
        const double pr = (t > 0.0 ? 0.0 : t) + (double)b;
        const double y = rcp_newton2(nrm);
#pragma unroll
        for (int i = 0; i < NC; i++) {
          const double num = pr * (double)av[i][j];
          odd |= __builtin_amdgcn_ballot_w64(!(f64_mid(num) || num == 0.0));
          out[i][j] = nz ? (T)div_mid(num, nrm, y) : (T)0;
        }
      }
      if (__builtin_expect(odd == 0, 1)) return;
    }
    // general expansions, still branch-free
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const bool nz = nv[j] > 0;
      const T nrm = nz ? t_sqrt(nv[j]) : (T)1;
      const T t = nrm - b;
      const T pr = (t > (T)0 ? (T)0 : t) + b;
#pragma unroll
      for (int i = 0; i < NC; i++) { const T q = pr * av[i][j] / nrm; out[i][j] = nz ? q : (T)0; }
    }
  }
}
constexpr float kTinyIsZeroRadius = 9.5367431640625e-07f;     // 2^-20

// fn is wave-uniform: a scalar branch, or resolved at compile time when FN >= 0
template <class T, int FN = -1>
__device__ __forceinline__ T f1d_apply(int fn_rt, T x0, T tau, T alpha, T beta) {
  const int fn = FN >= 0 ? FN : fn_rt;
  switch (fn) {
    case PROST_FN_ZERO: return x0;                                                   // :34-44
    case PROST_FN_ABS: return f1d_abs(x0, tau);
    case PROST_FN_SQUARE: return f1d_square(x0, tau);
    case PROST_FN_IND_LEQ0: return f1d_ind_leq0(x0);
    case PROST_FN_IND_GEQ0: return f1d_ind_geq0(x0);
    case PROST_FN_IND_EQ0: return (T)0;                                               // :105-114
    case PROST_FN_IND_BOX01: return f1d_ind_box01(x0);
    case PROST_FN_MAX_POS0: return f1d_max_pos0(x0, tau);
    case PROST_FN_L0: return f1d_l0(x0, tau);
    case PROST_FN_HUBER: return f1d_huber(x0, tau, alpha);
    case PROST_FN_LQ: return f1d_lq(x0, tau, alpha, beta);
    case PROST_FN_LQ_PLUS_EPS: return (T)0;                                           // :294-306
    case PROST_FN_TRUNCQUAD: return f1d_truncquad(x0, tau, alpha, beta);
    case PROST_FN_TRUNCLIN: return f1d_trunclin(x0, tau, alpha, beta);
  }
  return x0;
}

// step size of one element: invert_tau ? 1/(tau*td) : tau*td   (elem_operation_1d.hpp:40)
template <class T> __device__ __forceinline__ T elem_tau(T tau_scal, T td, bool invert_tau) {
  return invert_tau ? (T)(1. / (double)(tau_scal * td)) : (tau_scal * td);
}

// the scalar "c f(a x - b) + d x + e/2 x^2" prox on a value v (x itself for 1d, ||x|| for norm2)
// elem_operation_1d.hpp:45-57 == elem_operation_norm2.hpp:64-74
template <class T, int FN = -1>
__device__ __forceinline__ T scaled_prox(int fn, T v, T tau, const T* c) {
  const double den = 1. + (double)(tau * c[4]);
  const T prox_arg = (T)(div1((double)(c[0] * (v - c[3] * tau)), den) - (double)c[1]);
  const T step = (T)div1((double)(c[2] * c[0] * c[0] * tau), den);
  return div1((T)(f1d_apply<T, FN>(fn, prox_arg, step, c[5], c[6]) + c[1]), c[0]);
}

// ---- scaled prox with ALL seven coefficients and the step size wave-uniform --------------------
// Everything that does not depend on the element (the fp64 denominator 1. + tau*e, the step, the
// divisor of Function1DSquare and its reciprocal) is evaluated ONCE ON THE HOST with the same
// expression order, and the "is this divisor exactly 1" decisions become kernel-argument booleans
// (scalar branches) instead of per-lane selects around a 35-instruction fp64 division.
template <class T>
struct UniformProx {
  T c[7];
  T tau;
  double den;          // 1. + (double)(tau * c[4])
  T step;              // (T)((double)(c2*c0*c0*tau) / den)
  UniformDiv sq;       // 1. + (double)step and its reciprocal (Function1DSquare)
  bool den_one, a_one, degenerate;   // den == 1 ; c0 == 1 ; c0 == 0 || c2 == 0 (elem_operation_1d.hpp:42-44)
};
template <class T>
__host__ __device__ inline UniformProx<T> make_uniform_prox(const T* c, T tau) {
  UniformProx<T> u;
  for (int k = 0; k < 7; k++) u.c[k] = c[k];
  u.tau = tau;
  u.den = 1. + (double)(tau * c[4]);
  u.den_one = u.den == 1.0;
  const double num = (double)(c[2] * c[0] * c[0] * tau);
  u.step = (T)(u.den_one ? num : num / u.den);
  u.sq.D = 1. + (double)u.step;
  u.sq.rD = 1.0 / u.sq.D;
  u.a_one = c[0] == (T)1;
  u.degenerate = c[0] == (T)0 || c[2] == (T)0;
  return u;
}
// == scaled_prox<T, FN>(fn, v, u.tau, c) bit for bit.  c = this element's coefficients; c[0], c[2],
// c[4] (a, c, e) must be the uniform ones `u` was built from, the others (b, d, alpha, beta) may
// vary per element.
template <class T, int FN = -1>
__device__ __forceinline__ T scaled_prox_u(int fn, T v, const T* c, const UniformProx<T>& u) {
  const T num = c[0] * (v - c[3] * u.tau);
  T prox_arg;
  // den == 1: (T)((double)num - (double)b) == num - b in T.  For T = float this is Figueroa's theorem
  // (double rounding through a format of >= 2p+2 = 50 bits is innocuous for +, -, *, /, sqrt of
  // p = 24-bit operands; double has 53), so no fp64 instruction is needed.
  if (u.den_one) prox_arg = num - c[1];
  else prox_arg = (T)((double)num / u.den - (double)c[1]);
  T r;
  if ((FN >= 0 ? FN : fn) == PROST_FN_SQUARE) r = div_to_float_exact(prox_arg, u.sq);
  else r = f1d_apply<T, FN>(fn, prox_arg, u.step, c[5], c[6]);
  const T s = (T)(r + c[1]);
  return u.a_one ? s : s / c[0];
}
// == elem_1d<T, FN>(fn, arg, u.tau, c) bit for bit (same requirement on c)
template <class T, int FN = -1>
__device__ __forceinline__ T elem_1d_u(int fn, T arg, const T* c, const UniformProx<T>& u) {
  if (u.degenerate) return (arg - u.tau * c[3]) / (1 + u.tau * c[4]);
  return scaled_prox_u<T, FN>(fn, arg, c, u);
}

// ---- prox_f* as the Moreau wrap of a norm2 operation, straight-line ---------------------------
// prox_moreau.cu:98-134 with the dual call's invert_tau = false, around ElemOperationNorm2<FN> whose coefficients are all
// wave-uniform (elem_operation_norm2.hpp:40-88):   v = arg / s   (s = sigma Sigma; the caller forms it with SharedDivisor: the
// correctly rounded quotient),   r = pr v / ||v|| with pr = the scaled prox of ||v|| at the step 1 / s (`u` holds its terms),
// r = 0 where ||v|| = 0,   out = arg - s r.   nv = the squared norms of v as the reference accumulates them.  Norms in
// [2^-96, 2^126] take the short correctly rounded square root; the quotient runs through one refined double reciprocal per pixel
// (device_math.hpp: rcp_refined -- RN(n / d) for float n, d).  Same bits as the three expressions of the reference.
template <class T, int FN, int NC, int VEC>
__device__ __forceinline__ void norm2_moreau_post(const T (&nv)[VEC], const T (&vv)[NC][VEC], const T (&av)[NC][VEC], T s, const T* c, const UniformProx<T>& u,
                                                  T (&out)[NC][VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    const bool nz = nv[j] > (T)0;
    const T x = nz ? nv[j] : (T)1;
    T nrm;
    if constexpr (sizeof(T) == 4) {
      const bool mid = x >= (T)1.2621774483536189e-29f && x <= (T)8.507059173023462e37f;       // [2^-96, 2^126]
      nrm = __builtin_expect(mid, 1) ? sqrt_midrange(x) : t_sqrt(x);
    } else nrm = f64_mid((double)x) ? (T)sqrt_midrange((double)x) : t_sqrt(x);
    const T pr = scaled_prox_u<T, FN>(FN, nrm, c, u);
    if constexpr (sizeof(T) == 4) {
      const double r = rcp_refined(nrm);
#pragma unroll
      for (int i = 0; i < NC; i++) { const T q = nz ? (T)mul_rcp((float)(pr * vv[i][j]), r) : (T)0; out[i][j] = av[i][j] - s * q; }
    } else {
      // (fp64: the short reciprocal form where the operands allow it -- per lane here: this instance is not the hot one)
      const double y = rcp_newton2((double)nrm);
#pragma unroll
      for (int i = 0; i < NC; i++) {
        const double num = (double)(pr * vv[i][j]);
        const T qd = (f64_mid((double)nrm) && (f64_mid(num) || num == 0.0)) ? (T)div_mid(num, (double)nrm, y) : pr * vv[i][j] / nrm;
        const T q = nz ? qd : (T)0;
        out[i][j] = av[i][j] - s * q;
      }
    }
  }
}

// ---- scaled prox with two host-decided shortcuts (generic kernels, per-element step sizes) -----
// e_zero: coefficient e is the scalar 0  -> the fp64 denominator 1. + tau*e is exactly 1
// a_one : coefficient a is the scalar 1  -> the final division by a is the identity
// Both are kernel-argument booleans (scalar branches); with them false this is scaled_prox.
template <class T, int FN = -1>
__device__ __forceinline__ T scaled_prox_flags(int fn, T v, T tau, const T* c, bool e_zero, bool a_one) {
  const T num = c[0] * (v - c[3] * tau);
  const T snum = c[2] * c[0] * c[0] * tau;
  T prox_arg, step;
  if (e_zero) {
    prox_arg = num - c[1];                         // == (T)((double)num - (double)b), see scaled_prox_u
    step = snum;                                   // (T)((double)snum / 1.) == snum
  } else {
    const double den = 1. + (double)(tau * c[4]);
    prox_arg = (T)((double)num / den - (double)c[1]);
    step = (T)((double)snum / den);
  }
  const T s = (T)(f1d_apply<T, FN>(fn, prox_arg, step, c[5], c[6]) + c[1]);
  return a_one ? s : s / c[0];
}
template <class T, int FN = -1>
__device__ __forceinline__ T elem_1d_flags(int fn, T arg, T tau, const T* c, bool e_zero, bool a_one) {
  if (c[0] == 0 || c[2] == 0) return (arg - tau * c[3]) / (1 + tau * c[4]);
  return scaled_prox_flags<T, FN>(fn, arg, tau, c, e_zero, a_one);
}

// ElemOperation1D::operator() on one value (elem_operation_1d.hpp:36-59)
template <class T, int FN = -1>
__device__ __forceinline__ T elem_1d(int fn, T arg, T tau, const T* c) {
  if (c[0] == 0 || c[2] == 0) return (arg - tau * c[3]) / (1 + tau * c[4]);
  return scaled_prox<T, FN>(fn, arg, tau, c);
}

}  // namespace prost_hip
