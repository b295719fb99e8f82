// elementwise.hpp -- streaming map / map-reduce skeletons of the generic (unfused) kernels.
//
// A lane owns 16 bytes of consecutive elements per operand (float4 / double2): every access of a
// wave is one contiguous 1-KiB transaction, which is what saturates HBM3E on gfx950 (measured at
// 8192^2 floats: 16 B/lane 6.6 TB/s vs 4 B/lane 5.3 TB/s).  Operands that are not 16-byte aligned
// run the same kernel with VEC = 1; the n % VEC tail is done by the first lanes of workgroup 0.
// The per-element functor sees the NIN input values of ONE element, so the arithmetic (and its
// rounding) is exactly that of the scalar formula it was written from.
#pragma once
#include "fused_common.hpp"
#include "reduce.hpp"

namespace prost_hip {

template <class T, int NIN>
struct EwIn {
  const T* p[NIN > 0 ? NIN : 1];
};

// Functors whose scalars may live in a device-resident step-size record (prost_hip_use_step_record; kernels_pdhg.hip) define
// prepare() -- called once per thread before the first element: fetches the scalars (wave-uniform loads) -- and map kernels
// additionally skip(): true makes the whole launch a no-op (the record's stop flag).  Other functors have neither.
template <class F> __device__ __forceinline__ auto ew_prepare(F& f, int) -> decltype(f.prepare(), void()) { f.prepare(); }
template <class F> __device__ __forceinline__ void ew_prepare(F&, long) {}
template <class F> __device__ __forceinline__ auto ew_skip(const F& f, int) -> decltype(f.skip()) { return f.skip(); }
template <class F> __device__ __forceinline__ bool ew_skip(const F&, long) { return false; }

// out[i] = f(in[0][i], ..., in[NIN-1][i]); `out` may alias an input (all loads of an element
// group precede its store)
template <class T, int VEC, int NIN, class F>
__global__ void __launch_bounds__(kBlock) ew_kernel(T* out, EwIn<T, NIN> in, size_t n, F f) {
  if (ew_skip(f, 0)) return;
  ew_prepare(f, 0);
  const size_t nv = n / VEC;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (size_t)gridDim.x * kBlock) {
    T v[NIN > 0 ? NIN : 1][VEC], o[VEC];
#pragma unroll
    for (int k = 0; k < NIN; k++) ldv<T, VEC>(in.p[k] + i * VEC, v[k]);
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      T a[NIN > 0 ? NIN : 1];
#pragma unroll
      for (int k = 0; k < NIN; k++) a[k] = v[k][j];
      o[j] = f(a);
    }
    stv<T, VEC>(out + i * VEC, o);
  }
  if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < n - nv * VEC) {
    const size_t i = nv * VEC + threadIdx.x;
    T a[NIN > 0 ? NIN : 1];
#pragma unroll
    for (int k = 0; k < NIN; k++) a[k] = in.p[k][i];
    out[i] = f(a);
  }
}

template <class T, int NIN, class F>
static int launch_ew(const char* name, T* out, const EwIn<T, NIN>& in, size_t n, F f, hipStream_t st) {
  if (n == 0) return 0;
  constexpr int V = VecOf<T>::N;
  bool vec = aligned16(out) && n >= (size_t)V;
  for (int k = 0; k < NIN; k++) vec = vec && aligned16(in.p[k]);
  if (vec) hipLaunchKernelGGL((ew_kernel<T, V, NIN, F>), dim3(grid_for(n / V)), dim3(kBlock), 0, st, out, in, n, f);
  else hipLaunchKernelGGL((ew_kernel<T, 1, NIN, F>), dim3(grid_for(n)), dim3(kBlock), 0, st, out, in, n, f);
  PH_LAUNCH_END(name);
}

// (sum_a, sum_b) += f(element) in double; one partial pair per workgroup (reduce.hpp), folded by
// launch_fold.  The association order is fixed by (grid, VEC), hence run-to-run deterministic.
// the reduction of workgroup `block` of `nblocks` (its partial pair goes to slot `block`)
template <class T, int VEC, int NIN, class F>
__device__ __forceinline__ void reduce2_body(double* __restrict__ partial, const EwIn<T, NIN>& in, size_t n, F& f, unsigned block, unsigned nblocks) {
  ew_prepare(f, 0);
  const size_t nv = n / VEC;
  double sa = 0, sb = 0;
  for (size_t i = (size_t)block * kBlock + threadIdx.x; i < nv; i += (size_t)nblocks * kBlock) {
    T v[NIN][VEC];
#pragma unroll
    for (int k = 0; k < NIN; k++) ldv<T, VEC>(in.p[k] + i * VEC, v[k]);
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      T a[NIN];
#pragma unroll
      for (int k = 0; k < NIN; k++) a[k] = v[k][j];
      f(a, sa, sb);
    }
  }
  if (VEC > 1 && block == 0 && threadIdx.x < n - nv * VEC) {
    const size_t i = nv * VEC + threadIdx.x;
    T a[NIN];
#pragma unroll
    for (int k = 0; k < NIN; k++) a[k] = in.p[k][i];
    f(a, sa, sb);
  }
  block_sum2_store(sa, sb, partial, block);
}
template <class T, int VEC, int NIN, class F>
__global__ void __launch_bounds__(kBlock) reduce2_kernel(double* __restrict__ partial, EwIn<T, NIN> in, size_t n, F f) {
  reduce2_body<T, VEC, NIN, F>(partial, in, n, f, blockIdx.x, gridDim.x);
}
// TWO reductions in one launch: workgroups [0, g1) run the first exactly as reduce2_kernel on a grid of g1 would, the others the second
// on a grid of gridDim.x - g1 -- the same partials, bit for bit, as two launches
template <class T, int V1, int V2, int N1, int N2, class F1, class F2>
__global__ void __launch_bounds__(kBlock) reduce2_pair_kernel(double* __restrict__ partial1, EwIn<T, N1> in1, size_t n1, F1 f1, unsigned g1,
                                                             double* __restrict__ partial2, EwIn<T, N2> in2, size_t n2, F2 f2) {
  if (blockIdx.x < g1) reduce2_body<T, V1, N1, F1>(partial1, in1, n1, f1, blockIdx.x, g1);
  else reduce2_body<T, V2, N2, F2>(partial2, in2, n2, f2, blockIdx.x - g1, gridDim.x - g1);
}

// returns the number of partial slots written (0 on launch failure is reported through *err)
// grid of a reduction launch and whether it runs the 16-byte form
template <class T, int NIN>
static unsigned reduce2_geometry(const EwIn<T, NIN>& in, size_t n, bool& vec) {
  constexpr int V = VecOf<T>::N;
  vec = n >= (size_t)V;
  for (int k = 0; k < NIN; k++) vec = vec && aligned16(in.p[k]);
  unsigned g = grid_for(vec ? n / V : n, 4);
  if (g > (unsigned)kReduceBlocks) g = kReduceBlocks;
  return g;
}
template <class T, int NIN, class F>
static unsigned launch_reduce2(double* partial, const EwIn<T, NIN>& in, size_t n, F f, hipStream_t st) {
  constexpr int V = VecOf<T>::N;
  bool vec;
  const unsigned g = reduce2_geometry<T, NIN>(in, n, vec);
  if (vec) hipLaunchKernelGGL((reduce2_kernel<T, V, NIN, F>), dim3(g), dim3(kBlock), 0, st, partial, in, n, f);
  else hipLaunchKernelGGL((reduce2_kernel<T, 1, NIN, F>), dim3(g), dim3(kBlock), 0, st, partial, in, n, f);
  return g;
}

}  // namespace prost_hip
