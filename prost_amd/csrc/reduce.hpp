// reduce.hpp -- deterministic two-stage sum reduction of (a, b) pairs.
//
// Stage 1 (inside the producing kernel): 64-lane wavefront shuffle reduction, one LDS slot per
// wave, wave 0 folds the slots and writes ONE pair per workgroup to workspace[blockIdx] -- no
// atomics, so the result does not depend on scheduling.  Stage 2: one workgroup folds the
// workgroup partials in a fixed order.  Sums are carried in double.
#pragma once
#include "common.hpp"

namespace prost_hip {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
  return v;
}

// call from every thread of a kBlock-thread workgroup; writes partial[2*slot], partial[2*slot+1]
__device__ __forceinline__ void block_sum2_store(double a, double b, double* __restrict__ partial, unsigned slot) {
  __shared__ double s_a[kBlock / kWave], s_b[kBlock / kWave];
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) { s_a[wave] = a; s_b[wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0, tb = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; w++) { ta += s_a[w]; tb += s_b[w]; }
    partial[2 * slot] = ta;
    partial[2 * slot + 1] = tb;
  }
}

// stage 2 (kernels_pdhg.hip): out[0] = sum partial[2i], out[1] = sum partial[2i+1], i < nslots;
// optional sqrt of out[0]
int launch_fold(double* out, const double* partial, unsigned nslots, bool sqrt_first, hipStream_t s);


// ---- order-independent sums (round 5) ---------------------------------------------------------------------------
// The CG scalars of the ADMM graph projection (cgls.hpp:152-170: thrust::transform_reduce in double) and the norms of
// Problem::normest are reductions whose ORDER the reference leaves to thrust / cuBLAS.  alpha, beta and the relative stopping
// test of cgls.hpp:326-360 are knife-edge functions of them: two kernels that add the same terms in another order end a solve
// in different rounds now and then (profiles/r04_fuzz.log).  These sums are therefore carried as unevaluated pairs hi + lo
// (Knuth's TwoSum: the rounding error of every addition is kept), per thread, through the wavefront / workgroup folds and
// through the final fold: the result is the exact sum of the terms rounded ONCE (to within 2^-100 of the sum -- it can depend on
// the order only where the exact sum lies that close to a rounding boundary), so every kernel variant -- staged rounds, fused
// rounds, pixel-ordered rounds, any grid -- and the CPU oracle, which accumulates the same way, obtain the same double.
// Needs -ffp-contract=off (the whole library is built that way): no FMA may be formed from these additions.
struct dd_t { double hi, lo; };
__host__ __device__ __forceinline__ void dd_acc(dd_t& a, double t) {          // a += t
  const double s = a.hi + t;
  const double bb = s - a.hi;
  const double e = (a.hi - (s - bb)) + (t - bb);
  a.hi = s;
  a.lo += e;
}
__host__ __device__ __forceinline__ dd_t dd_add(dd_t a, dd_t b) {              // a + b, renormalised (|lo| <= ulp(hi) / 2)
  const double s = a.hi + b.hi;
  const double bb = s - a.hi;
  double e = (a.hi - (s - bb)) + (b.hi - bb);
  e += a.lo + b.lo;
  dd_t r;
  r.hi = s + e;
  r.lo = e - (r.hi - s);
  return r;
}
__device__ __forceinline__ dd_t wave_sum_dd(dd_t v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    dd_t w;
    w.hi = __shfl_down(v.hi, o, kWave);
    w.lo = __shfl_down(v.lo, o, kWave);
    v = dd_add(v, w);
  }
  return v;
}
// call from every thread of a kBlock-thread workgroup; partial[2 slot], partial[2 slot + 1] = (hi, lo) of the workgroup's sum
__device__ __forceinline__ void block_dd_store1(dd_t a, double* __restrict__ partial, unsigned slot) {
  __shared__ dd_t s_d1[kBlock / kWave];
  a = wave_sum_dd(a);
  if ((threadIdx.x & (kWave - 1)) == 0) s_d1[threadIdx.x / kWave] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    dd_t t = s_d1[0];
#pragma unroll
    for (int w = 1; w < kBlock / kWave; w++) t = dd_add(t, s_d1[w]);
    partial[2 * slot] = t.hi;
    partial[2 * slot + 1] = t.lo;
  }
  __syncthreads();                      // (s_d1 may be reused by a second call)
}
// two sums at once: partial[4 slot .. 4 slot + 3] = (a.hi, a.lo, b.hi, b.lo)
__device__ __forceinline__ void block_dd_store2(dd_t a, dd_t b, double* __restrict__ partial, unsigned slot) {
  __shared__ dd_t s_d2a[kBlock / kWave], s_d2b[kBlock / kWave];
  a = wave_sum_dd(a);
  b = wave_sum_dd(b);
  if ((threadIdx.x & (kWave - 1)) == 0) { s_d2a[threadIdx.x / kWave] = a; s_d2b[threadIdx.x / kWave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    dd_t ta = s_d2a[0], tb = s_d2b[0];
#pragma unroll
    for (int w = 1; w < kBlock / kWave; w++) { ta = dd_add(ta, s_d2a[w]); tb = dd_add(tb, s_d2b[w]); }
    partial[4 * slot] = ta.hi; partial[4 * slot + 1] = ta.lo;
    partial[4 * slot + 2] = tb.hi; partial[4 * slot + 3] = tb.lo;
  }
  __syncthreads();
}
// every thread of a kBlock-thread workgroup: the sum of g (hi, lo) partials, `stride` doubles apart (2: block_dd_store1 layout,
// 4: one of the two sums of block_dd_store2), rounded to double -- the same value in every workgroup of every kernel.
// Loads in batches of eight before the first add (a load-add loop pays one cache latency per iteration).
__device__ __forceinline__ double fold_dd(const double* __restrict__ part, unsigned g, unsigned stride) {
  __shared__ dd_t s_fd[kBlock / kWave];
  dd_t a{0.0, 0.0};
  for (unsigned base = threadIdx.x; base < g; base += 8 * kBlock) {
    double vh[8], vl[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const unsigned i = base + k * kBlock;
      vh[k] = i < g ? part[(size_t)stride * i] : 0.0;
      vl[k] = i < g ? part[(size_t)stride * i + 1] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) a = dd_add(a, dd_t{vh[k], vl[k]});
  }
  a = wave_sum_dd(a);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) s_fd[threadIdx.x / kWave] = a;
  __syncthreads();
  dd_t t = s_fd[0];
#pragma unroll
  for (int w = 1; w < kBlock / kWave; w++) t = dd_add(t, s_fd[w]);
  return t.hi;
}

// two folds at once (fold_dd): both sets of loads in flight together, one pair of barriers; every thread of every
// workgroup of every kernel obtains the same two doubles (order-independent sums)
__device__ __forceinline__ void fold_dd2(const double* __restrict__ pa, unsigned ga, unsigned sa_, const double* __restrict__ pb, unsigned gb, unsigned sb_,
                                         double& ra, double& rb) {
  __shared__ dd_t s_fa[kBlock / kWave], s_fb[kBlock / kWave];
  dd_t a{0.0, 0.0}, b{0.0, 0.0};
  const unsigned gmax = ga > gb ? ga : gb;
  for (unsigned base = threadIdx.x; base < gmax; base += 4 * kBlock) {
    double ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned i = base + k * kBlock;
      ah[k] = i < ga ? pa[(size_t)sa_ * i] : 0.0; al[k] = i < ga ? pa[(size_t)sa_ * i + 1] : 0.0;
      bh[k] = i < gb ? pb[(size_t)sb_ * i] : 0.0; bl[k] = i < gb ? pb[(size_t)sb_ * i + 1] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { a = dd_add(a, dd_t{ah[k], al[k]}); b = dd_add(b, dd_t{bh[k], bl[k]}); }
  }
  a = wave_sum_dd(a);
  b = wave_sum_dd(b);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) { s_fa[threadIdx.x / kWave] = a; s_fb[threadIdx.x / kWave] = b; }
  __syncthreads();
  dd_t ta = s_fa[0], tb = s_fb[0];
#pragma unroll
  for (int w = 1; w < kBlock / kWave; w++) { ta = dd_add(ta, s_fa[w]); tb = dd_add(tb, s_fb[w]); }
  ra = ta.hi; rb = tb.hi;
}

}  // namespace prost_hip
