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

}  // namespace prost_hip
