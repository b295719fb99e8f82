// kernels_normest_grad.hip -- one round of the operator-norm power iteration (Problem::normest, problem.cu:429-500) for an
// operator that is ONE gradient2d / gradient3d block under constant preconditioners, in ONE kernel.
//
// The generic round (kernels_cgls.hip: NORMEST_A, K, NORMEST_B, K^T, NORMEST_C) moves 23 values per voxel in 3-D (18 per pixel
// in 2-D) through HBM: x_temp, K x_temp and its rescaled copy are written and read back.  None of them is needed afterwards --
// the round maps x to x' = sqrt(T) K^T S K sqrt(T) (x / |x|) and reports two norms -- so here every thread forms, for its 4 rows
// of one column, the 7-point (5-point) stencil from x of the neighbouring rows / columns / planes in registers: 2 values per
// voxel through HBM (the neighbour loads are cache hits), 100 rounds at 2048 x 2048 x 64 in 0.06 s instead of 0.5 s.
// Every intermediate value is formed by the expression, in the precision and in the order the separate kernels use
// (x_temp = sqrt(tau) (x / |x|); a = sqrt(sigma) (K x_temp); sqrt(sigma) a; 0 - (divx + divy + divl); sqrt(tau) K^T ...), so x'
// is bit-identical to the generic round; the two norms are order-independent sums of the same terms (reduce.hpp, dd_t): the
// same doubles as the generic round's and the oracle's.
#include "fused_common.hpp"
#include "reduce.hpp"

namespace prost_hip {

template <class T, int VEC, bool D3>
__global__ void __launch_bounds__(kBlock) normest_grad_kernel(T* __restrict__ x_out, const T* __restrict__ x_in, size_t nx, size_t ny, size_t L,
                                                              unsigned strips, T tau, T sigma, const double* __restrict__ norm_from,
                                                              double* __restrict__ partial) {
  const T sqT = t_sqrt(tau), sqS = t_sqrt(sigma);
  const double nf = norm_from ? *norm_from : 0.0;
  const bool divide = nf != 0.0;                       // first round: no divide (NORMEST_A)
  const SharedDivisor<T> by_norm(divide ? (T)nf : (T)1);
  auto xt = [&](T v) { return sqT * (divide ? by_norm.div(v) : v); };          // x_temp of NORMEST_A
  const size_t slice = nx * ny;
  const size_t tiles = (size_t)strips * nx * L;
  dd_t sa{0.0, 0.0}, sx{0.0, 0.0};      // order-independent sums (reduce.hpp): the norms equal the staged round's and the oracle's
  for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const size_t strip = t % strips, c = (t / strips) % nx, l = t / ((size_t)strips * nx);
    const size_t row0 = (strip * kBlock + threadIdx.x) * VEC;
    if (row0 >= ny) continue;
    const T* col = x_in + l * slice + c * ny;
    T xc[VEC], xl[VEC], xr[VEC], xd[VEC], xu[VEC];       // column c, c-1, c+1, plane l-1, l+1
#pragma unroll
    for (int j = 0; j < VEC; j++) { xl[j] = 0; xr[j] = 0; xd[j] = 0; xu[j] = 0; }
    ldv<T, VEC>(col + row0, xc);
    if (c > 0) ldv<T, VEC>(col - ny + row0, xl);
    if (c + 1 < nx) ldv<T, VEC>(col + ny + row0, xr);
    if (D3 && l > 0) ldv<T, VEC>(col - slice + row0, xd);
    if (D3 && l + 1 < L) ldv<T, VEC>(col + slice + row0, xu);
    const T above = row0 > 0 ? xt(col[row0 - 1]) : (T)0;               // x_temp of the row above / below this lane's rows
    const T below = row0 + VEC < ny ? xt(col[row0 + VEC]) : (T)0;
#pragma unroll
    for (int j = 0; j < VEC; j++) { xc[j] = xt(xc[j]); xl[j] = xt(xl[j]); xr[j] = xt(xr[j]); if (D3) { xd[j] = xt(xd[j]); xu[j] = xt(xu[j]); } }
    T o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const size_t row = row0 + j;
      const T val = xc[j];
      const T up = j > 0 ? xc[j > 0 ? j - 1 : 0] : above, dn = j + 1 < VEC ? xc[j + 1 < VEC ? j + 1 : 0] : below;
      // K x_temp at this voxel and at its lower neighbours (grad_fwd_vec_kernel), then NORMEST_B: a = sqrt(sigma) k, sqrt(sigma) a
      const T gx = c + 1 < nx ? xr[j] - val : (T)0, gxm = xc[j] - xl[j];            // gxm: at column c-1 (used only when c > 0)
      const T gy = row < ny - 1 ? dn - val : (T)0, gym = val - up;                  // gym: at row-1 (used only when row > 0)
      const T ax = sqS * gx, axm = sqS * gxm, ay = sqS * gy, aym = sqS * gym;
      dd_acc(sa, (double)ax * (double)ax); dd_acc(sa, (double)ay * (double)ay);
      // K^T (grad_adj_vec_kernel): 0 - (divx + divy [+ divl])
      T divy = row < ny - 1 ? sqS * ay : (T)0;
      if (row > 0) divy -= sqS * aym;
      T divx = c < nx - 1 ? sqS * ax : (T)0;
      if (c > 0) divx -= sqS * axm;
      T s = divx + divy;
      if (D3) {
        const T gl = l + 1 < L ? xu[j] - val : -val, glm = val - xd[j];             // Dirichlet above the last plane; glm: plane l-1
        const T al = sqS * gl, alm = sqS * glm;
        dd_acc(sa, (double)al * (double)al);
        T divl = sqS * al;
        if (l > 0) divl -= sqS * alm;
        s = divx + divy + divl;
      }
      const T kty = (T)0 - s;
      o[j] = sqT * kty;                                                           // NORMEST_C
      dd_acc(sx, (double)o[j] * (double)o[j]);
    }
    stv<T, VEC>(x_out + l * slice + c * ny + row0, o);
  }
  block_dd_store2(sa, sx, partial, blockIdx.x);
}

// out[0] = sqrt(first sum), out[1] = sqrt(second sum) of g two-sum slots (block_dd_store2)
__global__ void __launch_bounds__(kBlock) normest_fold_kernel(double* out, const double* __restrict__ partial, unsigned g) {
  const double ta = fold_dd(partial, g, 4), tb = fold_dd(partial + 2, g, 4);
  if (threadIdx.x == 0) { out[0] = sqrt(ta); out[1] = sqrt(tb); }
}

template <class T>
static int normest_grad_round(const prost_hip_normest_grad_desc* d, void* stream) {
  if (!d || !d->x_in || !d->x_out || !d->out || !d->workspace) { set_error("normest gradient round: null argument"); return 1; }
  if (d->x_in == d->x_out) { set_error("normest gradient round: the output must not alias the input (stencil)"); return 1; }
  if (d->nx == 0 || d->ny == 0 || d->L == 0) { set_error("normest gradient round: empty operator"); return 1; }
  constexpr int V = VecOf<T>::N;
  const bool vec = d->ny % V == 0 && aligned16(d->x_in) && aligned16(d->x_out);
  const size_t rows_per_block = (size_t)kBlock * (vec ? V : 1);
  const size_t strips = (d->ny + rows_per_block - 1) / rows_per_block;
  const size_t tiles = strips * d->nx * d->L;
  if (strips > 0xFFFFFFFFull) { set_error("normest gradient round: too many rows"); return 1; }
  const unsigned grid = (unsigned)(tiles < (size_t)kReduceBlocks ? tiles : (size_t)kReduceBlocks);
  hipStream_t s = as_stream(stream);
  T* xo = static_cast<T*>(d->x_out); const T* xi = static_cast<const T*>(d->x_in);
  double* partial = static_cast<double*>(d->workspace);
#define GO(VV, D3v) hipLaunchKernelGGL((normest_grad_kernel<T, VV, D3v>), dim3(grid), dim3(kBlock), 0, s, xo, xi, (size_t)d->nx, (size_t)d->ny, (size_t)d->L, (unsigned)strips, (T)d->tau, (T)d->sigma, d->norm_x_from, partial)
  if (d->is3d) { if (vec) GO(V, true); else GO(1, true); }
  else { if (vec) GO(V, false); else GO(1, false); }
#undef GO
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "normest gradient round kernel"); }
  hipLaunchKernelGGL(normest_fold_kernel, dim3(1), dim3(kBlock), 0, s, d->out, partial, grid);
  PH_LAUNCH_END("normest fold kernel");
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_normest_grad_round_f32(const prost_hip_normest_grad_desc* d, void* stream) { return normest_grad_round<float>(d, stream); }
int prost_hip_normest_grad_round_f64(const prost_hip_normest_grad_desc* d, void* stream) { return normest_grad_round<double>(d, stream); }
}  // extern "C"
