// kernels_selftest.hip -- exhaustive-style self test of the short correctly rounded forms of
// device_math.hpp against the compiler's IEEE expansions, ON the device (test infrastructure of
// the kernel library itself: tests/test_gpu_kernels.py::test_short_division_and_sqrt_forms).
// The two-iterations kernel relies on  mul_rcp(n, rcp_refined(d)) == n / d,
// sqrt_midrange(x) == sqrtf(x) on [2^-96, 2^126],  div_to_float_exact(x, u) == (float)((double)x / u.D)
// and  a - b == (float)((double)a - (double)b)  for ALL inputs in their stated domains.
#include "common.hpp"
#include "device_math.hpp"

namespace prost_hip {

__device__ __forceinline__ uint32_t mix32(uint64_t v) {
  v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull; v ^= v >> 33;
  return (uint32_t)v;
}
// a float with a uniformly random mantissa and sign, exponent uniform in [elo, ehi] (unbiased)
__device__ __forceinline__ float rnd_float(uint64_t key, int elo, int ehi) {
  const uint32_t r = mix32(key), e = mix32(key ^ 0x9e3779b97f4a7c15ull);
  const int ex = elo + (int)(e % (uint32_t)(ehi - elo + 1));
  return __uint_as_float((r & 0x807FFFFFu) | ((uint32_t)(ex + 127) << 23));
}

__global__ void __launch_bounds__(kBlock) selftest_kernel(unsigned long long* bad, uint64_t n, uint64_t seed) {
  unsigned long long b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const uint64_t k = seed * 0x100000001b3ull + i * 4;
    // (0) division through the refined double reciprocal: any normal denominator, numerators over the
    //     whole float range including subnormal and overflowing quotients, zeros, and n == +-d
    const int sel = (int)(i % 5);
    float d = rnd_float(k, -126, 127);
    if (sel == 3) d = fabsf(d);
    float nn = sel == 0 ? rnd_float(k + 1, -126, 127) : sel == 1 ? rnd_float(k + 1, -149 + 23, -100) * 1e-10f : sel == 2 ? 0.0f : sel == 3 ? -d : rnd_float(k + 1, -20, 20) * d;
    const float q = mul_rcp(nn, rcp_refined(d)), qr = nn / d;
    b0 += (__float_as_uint(q) != __float_as_uint(qr)) && !(q != q && qr != qr);
    // the same product rounded through fma(n, r, +0): the quotient of a ZERO numerator becomes +0, everything else is
    // unchanged (a non-zero quotient that underflows to zero keeps its sign, as n / d does)
    const float qz = mul_rcp_pz(nn, rcp_refined(d)), qzr = nn == 0.0f ? nn / d + 0.0f : nn / d;
    b5 += (__float_as_uint(qz) != __float_as_uint(qzr)) && !(qz != qz && qzr != qzr);
    // min0(t) == (t > 0 ? 0 : t) for every t that is not NaN, signed zeros included
    const float t0 = sel == 2 ? (i & 8 ? -0.0f : 0.0f) : nn;
    b6 += __float_as_uint(min0(t0)) != __float_as_uint(t0 > 0.0f ? 0.0f : t0);
    // (4) control: the un-refined single-precision reciprocal is NOT exact -- this count must be > 0
    b4 += __float_as_uint(nn * __builtin_amdgcn_rcpf(d)) != __float_as_uint(qr);
    // (1) sqrt without range scaling on its whole domain [2^-96, 2^126]
    const float x = fabsf(rnd_float(k + 2, -96, 125));
    const float s = sqrt_midrange(x), sr = sqrtf(x);
    b1 += __float_as_uint(s) != __float_as_uint(sr);
    // (2) exact (float)((double)x / D), D = 1 + step as Function1DSquare forms it, steps from tiny to huge
    const float step = fabsf(rnd_float(k + 3, -30, 30));
    const UniformDiv u = make_uniform_div(1. + (double)step);
    const float xv = rnd_float(k + 1, -60, 60);
    float in4[1] = {xv}, out4[1];
    div_to_float_exact_vec<1>(in4, u, out4);
    const float er = (float)((double)xv / u.D);
    b2 += __float_as_uint(out4[0]) != __float_as_uint(er);
    b2 += __float_as_uint(div_to_float_exact(xv, u)) != __float_as_uint(er);
    // (7) fp64: the short quotient and square root of device_math.hpp == `/` and sqrt for operands in their stated range
    //     (|v| in [2^-500, 2^500], numerators also 0), the vector form and SharedDivisor over the WHOLE range (their guards)
    {
      const uint64_t m1 = ((uint64_t)mix32(k + 11) << 32) | mix32(k + 12), m2 = ((uint64_t)mix32(k + 13) << 32) | mix32(k + 14);
      const int full = (int)((i / 5) % 4) == 3;                      // one case in four: exponents over the whole double range
      const int e1 = full ? (int)(mix32(k + 15) % 2046u) + 1 : 1023 - 500 + (int)(mix32(k + 15) % 1001u);
      const int e2 = full ? (int)(mix32(k + 16) % 2046u) + 1 : 1023 - 500 + (int)(mix32(k + 16) % 1001u);
      double dn = __longlong_as_double((long long)((m1 & 0x800FFFFFFFFFFFFFull) | ((uint64_t)e1 << 52)));
      const double dd = __longlong_as_double((long long)((m2 & 0x800FFFFFFFFFFFFFull) | ((uint64_t)e2 << 52)));
      if (sel == 2) dn = 0.0; else if (sel == 3) dn = -dd;
      const double qr64 = dn / dd;
      if (!full) {
        const double q64 = div_mid(dn, dd, rcp_newton2(dd));
        b7 += (__double_as_longlong(q64) != __double_as_longlong(qr64)) && !(dn == 0.0 && q64 == 0.0 && qr64 == 0.0);
        const double ax = fabs(dd);
        b7 += __double_as_longlong(sqrt_midrange(ax)) != __double_as_longlong(sqrt(ax));
      }
      const SharedDivisor<double> sd(dd);
      const double qs = sd.div(dn);
      b7 += (__double_as_longlong(qs) != __double_as_longlong(qr64)) && !(qs == 0.0 && qr64 == 0.0) && !(qs != qs && qr64 != qr64);
      const UniformDiv u64 = make_uniform_div(fabs(dd));
      double in1[1] = {dn}, out1[1];
      div_to_float_exact_vec<1>(in1, u64, out1);
      const double qv = dn / u64.D;
      b7 += (__double_as_longlong(out1[0]) != __double_as_longlong(qv)) && !(out1[0] == 0.0 && qv == 0.0) && !(out1[0] != out1[0] && qv != qv);
      const double q1 = div_to_float_exact(dn, u64);
      b7 += (__double_as_longlong(q1) != __double_as_longlong(qv)) && !(q1 == 0.0 && qv == 0.0) && !(q1 != q1 && qv != qv);
    }
    // (3) float subtraction == double subtraction rounded to float (exponent gaps up to the full range)
    const float a = rnd_float(k + 1, -126, 127), c = rnd_float(k + 2, -126, 127);
    const float sf = a - c, sd = (float)((double)a - (double)c);
    b3 += __float_as_uint(sf) != __float_as_uint(sd);
  }
  if (b0) atomicAdd(bad + 0, b0);
  if (b1) atomicAdd(bad + 1, b1);
  if (b2) atomicAdd(bad + 2, b2);
  if (b3) atomicAdd(bad + 3, b3);
  if (b4) atomicAdd(bad + 4, b4);
  if (b5) atomicAdd(bad + 5, b5);
  if (b6) atomicAdd(bad + 6, b6);
  if (b7) atomicAdd(bad + 7, b7);
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" int prost_hip_selftest_math(unsigned long long* mismatches8, uint64_t n, uint64_t seed, void* stream) {
  PH_CHECK(hipMemsetAsync(mismatches8, 0, 8 * sizeof(unsigned long long), as_stream(stream)));
  if (n == 0) return 0;
  hipLaunchKernelGGL(selftest_kernel, dim3(4096), dim3(kBlock), 0, as_stream(stream), mismatches8, n, seed);
  PH_LAUNCH_END("selftest kernel");
}
