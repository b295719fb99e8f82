// kernels_fused3d.hip -- fused PDHG passes for gradient3d problems (volumetric TV, BASELINE config 3).
//
// Same mapping as kernels_fused.hip (16-byte lanes along y, workgroups marching over column
// chunks with the x-neighbour column kept in registers, y-neighbour by wave shuffle); the third
// difference runs along l with stride nx*ny (block_gradient3d.cu:57-60), Dirichlet at l = L-1
// (:73-76).  grid.z = l: workgroups of consecutive planes are dispatched back to back, so the
// plane l+-1 column a workgroup reads for the l-difference is the one its neighbour in z streams
// as its own centre plane at about the same time -- served from L2 / Infinity Cache (a 2048^2
// fp32 plane is 16 MiB, the Infinity Cache holds 256 MiB).
// Algorithmic traffic: primal reads y (3n), x (n), f (n), writes x (n); dual reads
// y (3n), x_new (n), x_old (n), writes y (3n)  = 14 floats / voxel / iteration.
#include "fused_common.hpp"
#include "reduce.hpp"

namespace prost_hip {

// ------------------------------------------------------------------------------------------
// primal pass: x_new = prox_g(x - tau T K^T y)   [+ dual residual sums]
// ------------------------------------------------------------------------------------------
template <class T, int VEC, int GFN, bool RES>
__global__ void __launch_bounds__(kBlock) fused_primal3d_kernel(T* __restrict__ x_new, const T* __restrict__ x,
                                                                const T* __restrict__ y, const T* __restrict__ y_prev,
                                                                FusedArgs<T> a, UniformProx<T> ug, bool g_uniform, T tau,
                                                                bool use_kty, bool use_kty_prev, double* __restrict__ partial) {
  const size_t nx = a.nx, ny = a.ny, L = a.L;
  const size_t row0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * VEC;
  const bool active = row0 < ny;
  const size_t xa = (size_t)blockIdx.y * a.cols_per_block;
  const size_t xb = xa + a.cols_per_block < nx ? xa + a.cols_per_block : nx;
  const size_t l = blockIdx.z;
  const size_t P = nx * ny, N = P * L, plane = l * P;
  const T* y1 = y + plane;             // d/dx   (res[idx],        block_gradient3d.cu:78)
  const T* y2 = y + N + plane;         // d/dy   (res[idx + N],    :79)
  const T* y3 = y + 2 * N + plane;     // d/dl   (res[idx + 2N],   :80)
  const T* p1 = RES ? y_prev + plane : nullptr;
  const T* p2 = RES ? y_prev + N + plane : nullptr;
  const T* p3 = RES ? y_prev + 2 * N + plane : nullptr;
  const T tauT = tau * a.Tval;
  const T sqT = t_sqrt(a.Tval);
  double ra = 0, rb = 0;

  T y1p[VEC], q1p[VEC];
#pragma unroll
  for (int j = 0; j < VEC; j++) { y1p[j] = 0; q1p[j] = 0; }
  if (active && xa > 0) {
    ldv<T, VEC>(y1 + (xa - 1) * ny + row0, y1p);
    if (RES) ldv<T, VEC>(p1 + (xa - 1) * ny + row0, q1p);
  }
  for (size_t xc = xa; xc < xb; xc++) {
    const size_t cb = xc * ny;
    T y1c[VEC], y2c[VEC], y3c[VEC], y3m[VEC], xv[VEC], q1c[VEC], q2c[VEC], q3c[VEC], q3m[VEC], gc[7][VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) { y1c[j] = 0; y2c[j] = 0; y3c[j] = 0; y3m[j] = 0; xv[j] = 0; q1c[j] = 0; q2c[j] = 0; q3c[j] = 0; q3m[j] = 0; }
    if (active) {
      ldv<T, VEC>(y1 + cb + row0, y1c);
      ldv<T, VEC>(y2 + cb + row0, y2c);
      ldv<T, VEC>(y3 + cb + row0, y3c);
      if (l > 0) ldv<T, VEC>(y3 - P + cb + row0, y3m);                 // plane l-1 (block_gradient3d.cu:144-147)
      ldv<T, VEC>(x + plane + cb + row0, xv);
      if (RES) {
        ldv<T, VEC>(p1 + cb + row0, q1c); ldv<T, VEC>(p2 + cb + row0, q2c); ldv<T, VEC>(p3 + cb + row0, q3c);
        if (l > 0) ldv<T, VEC>(p3 - P + cb + row0, q3m);
      }
#pragma unroll
      for (int k = 0; k < 7; k++) {
        if (a.g_ptr[k]) ldv<T, VEC>(a.g_ptr[k] + plane + cb + row0, gc[k]);
        else {
#pragma unroll
          for (int j = 0; j < VEC; j++) gc[k][j] = a.g_val[k];
        }
      }
    }
    const T up = row_above<T, VEC>(y2c, y2 + cb, row0, active);
    T upp = 0;
    if (RES) upp = row_above<T, VEC>(q2c, p2 + cb, row0, active);
    if (active) {
      T out[VEC];
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const size_t row = row0 + j;
        // BlockGradient3DKernelAdjoint (block_gradient3d.cu:127-149) on a zero-filled result
        T divy = (row < ny - 1) ? y2c[j] : (T)0;
        if (row > 0) divy -= (j > 0 ? y2c[j > 0 ? j - 1 : 0] : up);
        T divx = (xc < nx - 1) ? y1c[j] : (T)0;
        if (xc > 0) divx -= y1p[j];
        T divl = y3c[j];
        if (l > 0) divl -= y3m[j];
        const T kty = use_kty ? (T)0 - (divx + divy + divl) : (T)0;
        const T arg = xv[j] - tauT * kty;
        T c[7];
#pragma unroll
        for (int k = 0; k < 7; k++) c[k] = gc[k][j];
        out[j] = g_uniform ? elem_1d_u<T, GFN>(a.g_fn, arg, c, ug) : elem_1d<T, GFN>(a.g_fn, arg, tauT, c);
        if (RES) {
          T dpy = (row < ny - 1) ? q2c[j] : (T)0;
          if (row > 0) dpy -= (j > 0 ? q2c[j > 0 ? j - 1 : 0] : upp);
          T dpx = (xc < nx - 1) ? q1c[j] : (T)0;
          if (xc > 0) dpx -= q1p[j];
          T dpl = q3c[j];
          if (l > 0) dpl -= q3m[j];
          const T ktyp = use_kty_prev ? (T)0 - (dpx + dpy + dpl) : (T)0;
          const T w_hat = (xv[j] - out[j]) / (tau * sqT) - sqT * ktyp;     // backend_pdhg.cu:73-94
          const T diff = w_hat + sqT * kty;
          ra += (double)(diff * diff);
          rb += (double)(w_hat * w_hat);
        }
      }
      stv_nt<T, VEC>(x_new + plane + cb + row0, out);
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) { y1p[j] = y1c[j]; if (RES) q1p[j] = q1c[j]; }
  }
  if (RES) block_sum2_store(ra, rb, partial, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// ------------------------------------------------------------------------------------------
// dual pass: y_new = prox_f*(y + sigma S ((1+theta) K x_new - theta K x_old)), 3 components / voxel
// ------------------------------------------------------------------------------------------
template <class T, int VEC, int FFN, bool RES>
__global__ void __launch_bounds__(kBlock) fused_dual3d_kernel(T* __restrict__ y_new, const T* __restrict__ y,
                                                              const T* __restrict__ xn, const T* __restrict__ xo, FusedArgs<T> a,
                                                              UniformProx<T> uf, bool f_uniform, T sigma, T theta, bool use_kx_prev,
                                                              double* __restrict__ partial) {
  const size_t nx = a.nx, ny = a.ny, L = a.L;
  const size_t row0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * VEC;
  const bool active = row0 < ny;
  const size_t xa = (size_t)blockIdx.y * a.cols_per_block;
  const size_t xb = xa + a.cols_per_block < nx ? xa + a.cols_per_block : nx;
  const size_t l = blockIdx.z;
  const size_t P = nx * ny, N = P * L, plane = l * P;
  const bool has_above = l + 1 < L;
  const T sigS = sigma * a.Sval;
  const T sqS = t_sqrt(a.Sval);
  double ra = 0, rb = 0;

  T cn[VEC], co[VEC];
#pragma unroll
  for (int j = 0; j < VEC; j++) { cn[j] = 0; co[j] = 0; }
  if (active) { ldv<T, VEC>(xn + plane + xa * ny + row0, cn); ldv<T, VEC>(xo + plane + xa * ny + row0, co); }
  for (size_t xc = xa; xc < xb; xc++) {
    const size_t cb = xc * ny;
    const bool has_next = xc < nx - 1;
    T nn[VEC], no[VEC], an[VEC], ao[VEC], ya[3][VEC], fc[7][VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) { nn[j] = 0; no[j] = 0; an[j] = 0; ao[j] = 0; ya[0][j] = 0; ya[1][j] = 0; ya[2][j] = 0; }
    if (active) {
      if (has_next) { ldv<T, VEC>(xn + plane + cb + ny + row0, nn); ldv<T, VEC>(xo + plane + cb + ny + row0, no); }
      if (has_above) { ldv<T, VEC>(xn + plane + P + cb + row0, an); ldv<T, VEC>(xo + plane + P + cb + row0, ao); }   // plane l+1
      ldv<T, VEC>(y + plane + cb + row0, ya[0]);
      ldv<T, VEC>(y + N + plane + cb + row0, ya[1]);
      ldv<T, VEC>(y + 2 * N + plane + cb + row0, ya[2]);
#pragma unroll
      for (int k = 0; k < 7; k++) {
        if (a.f_ptr[k]) ldv<T, VEC>(a.f_ptr[k] + plane + cb + row0, fc[k]);
        else {
#pragma unroll
          for (int j = 0; j < VEC; j++) fc[k][j] = a.f_val[k];
        }
      }
    }
    const T dnn = row_below<T, VEC>(cn, xn + plane + cb, row0, ny, active);
    const T dno = row_below<T, VEC>(co, xo + plane + cb, row0, ny, active);
    if (active) {
      T out[3][VEC];
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const size_t row = row0 + j;
        // BlockGradient3DKernel (block_gradient3d.cu:62-80) on a zero-filled result
        const T below_n = (j < VEC - 1) ? cn[j < VEC - 1 ? j + 1 : 0] : dnn;
        const T below_o = (j < VEC - 1) ? co[j < VEC - 1 ? j + 1 : 0] : dno;
        T kx[3], kp[3], arg[3];
        kx[0] = has_next ? nn[j] - cn[j] : (T)0;
        kx[1] = (row < ny - 1) ? below_n - cn[j] : (T)0;
        kx[2] = has_above ? an[j] - cn[j] : -cn[j];                           // Dirichlet (:73-76)
        kp[0] = (use_kx_prev && has_next) ? no[j] - co[j] : (T)0;
        kp[1] = (use_kx_prev && row < ny - 1) ? below_o - co[j] : (T)0;
        kp[2] = use_kx_prev ? (has_above ? ao[j] - co[j] : -co[j]) : (T)0;
        T norm = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          arg[i] = ya[i][j] + sigS * ((1 + theta) * kx[i] - theta * kp[i]);    // backend_pdhg.cu:54-70
          norm += arg[i] * arg[i];
        }
        if (norm > 0) {
          norm = t_sqrt(norm);
          T c[7];
#pragma unroll
          for (int k = 0; k < 7; k++) c[k] = fc[k][j];
          const T pr = f_uniform ? scaled_prox_u<T, FFN>(a.f_fn, norm, c, uf) : scaled_prox<T, FFN>(a.f_fn, norm, sigS, c);
#pragma unroll
          for (int i = 0; i < 3; i++) out[i][j] = pr * arg[i] / norm;
        } else {
#pragma unroll
          for (int i = 0; i < 3; i++) out[i][j] = 0;
        }
        if (RES) {                                                             // backend_pdhg.cu:97-120
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const T z_hat = (ya[i][j] - out[i][j]) / (sigma * sqS) + sqS * ((1 + theta) * kx[i] - theta * kp[i]);
            const T diff = z_hat - sqS * kx[i];
            ra += (double)(diff * diff);
            rb += (double)(z_hat * z_hat);
          }
        }
      }
      stv_nt<T, VEC>(y_new + plane + cb + row0, out[0]);
      stv_nt<T, VEC>(y_new + N + plane + cb + row0, out[1]);
      stv_nt<T, VEC>(y_new + 2 * N + plane + cb + row0, out[2]);
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) { cn[j] = nn[j]; co[j] = no[j]; }
  }
  if (RES) block_sum2_store(ra, rb, partial, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// ------------------------------------------------------------------------------------------
bool fused3d_desc_ok(const prost_hip_fused_desc* d) {
  if (!d || !d->is3d || d->var_T || d->f_moreau) return false;
  if (d->nx == 0 || d->ny == 0 || d->L == 0 || d->L > 65535) return false;
  if (d->g_fn < 0 || d->g_fn >= PROST_FN_COUNT || d->f_fn < 0 || d->f_fn >= PROST_FN_COUNT) return false;
  const size_t rb = (d->ny + kBlock - 1) / kBlock;        // worst case VEC = 1
  // residual launches keep one partial per workgroup: even with one column chunk (the widest pick_cols3d can choose) the grid
  // of row blocks x planes must fit the reduction workspace -- larger volumes take the generic path
  if (rb * d->L > (size_t)kReduceBlocks) return false;
  return rb <= 65535 && d->nx <= 65535 * 4;
}

template <class T>
static bool vec3_ok(const prost_hip_fused_desc* d, const void* p0, const void* p1, const void* p2, const void* p3) {
  if (d->ny % VecOf<T>::N != 0) return false;
  bool ok = aligned16(p0) && aligned16(p1) && aligned16(p2) && aligned16(p3);
  for (int k = 0; k < 7; k++) ok = ok && aligned16(d->g_coeff_ptr[k]) && aligned16(d->f_coeff_ptr[k]);
  return ok;
}

// column chunk: off powers of two (HBM channel collisions, see kernels_fused_iter.hip), grid large
// enough to fill the chip, residual grids bounded by the reduction workspace
static int pick_cols3d(size_t nx, size_t row_blocks, size_t planes, bool res) {
  size_t cols = 12;
  while (cols > 6 && row_blocks * planes * ((nx + cols - 1) / cols) < 2048) cols -= 3;
  while (res && cols < nx && row_blocks * planes * ((nx + cols - 1) / cols) > (size_t)kReduceBlocks) cols += 6;      // ends at one chunk
  return (int)cols;
}

template <class T>
int run_primal3d(const prost_hip_fused_desc* d, T* x_new, const T* x, const T* y, const T* y_prev, double tau, int use_kty,
                 int use_kty_prev, double* out2, void* ws, void* stream) {
  if (!fused3d_desc_ok(d)) { set_error("fused 3-D primal pass: unsupported description"); return 1; }
  if (out2 && (!ws || !y_prev)) { set_error("fused 3-D primal pass: residuals need workspace and y_prev"); return 1; }
  FusedArgs<T> a = make_fused_args<T>(d);
  const bool vec = vec3_ok<T>(d, x_new, x, y, out2 ? y_prev : nullptr);
  const int V = vec ? VecOf<T>::N : 1;
  const size_t rb = (d->ny + (size_t)kBlock * V - 1) / ((size_t)kBlock * V);
  a.cols_per_block = pick_cols3d(d->nx, rb, d->L, out2 != nullptr);
  dim3 grid((unsigned)rb, (unsigned)((d->nx + a.cols_per_block - 1) / a.cols_per_block), (unsigned)d->L), block(kBlock);
  if (out2 && (size_t)grid.x * grid.y * grid.z > (size_t)kReduceBlocks) { set_error("fused 3-D primal pass: grid exceeds the reduction workspace"); return 1; }
  hipStream_t s = as_stream(stream);
  double* partial = static_cast<double*>(ws);
  const UniformProx<T> ug = make_uniform_prox<T>(a.g_val, (T)tau * a.Tval);
  const bool g_uniform = !d->g_coeff_ptr[0] && !d->g_coeff_ptr[2] && !d->g_coeff_ptr[4];
  const bool fast = d->g_fn == PROST_FN_SQUARE;
#define GO(VECv, GFNv, RESv) PH_LAUNCH((fused_primal3d_kernel<T, VECv, GFNv, RESv>), grid, block, 0, s, x_new, x, y, y_prev, a, ug, g_uniform, (T)tau, use_kty != 0, use_kty_prev != 0, partial)
  if (vec) {
    if (fast) { if (out2) GO(VecOf<T>::N, PROST_FN_SQUARE, true); else GO(VecOf<T>::N, PROST_FN_SQUARE, false); }
    else { if (out2) GO(VecOf<T>::N, -1, true); else GO(VecOf<T>::N, -1, false); }
  } else {
    if (out2) GO(1, -1, true); else GO(1, -1, false);
  }
#undef GO
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "fused 3-D primal pass"); }
  if (out2) return launch_fold(out2, partial, grid.x * grid.y * grid.z, false, s);
  return 0;
}

template <class T>
int run_dual3d(const prost_hip_fused_desc* d, T* y_new, const T* y, const T* xn, const T* xo, double sigma, double theta,
               int use_kx_prev, double* out2, void* ws, void* stream) {
  if (!fused3d_desc_ok(d)) { set_error("fused 3-D dual pass: unsupported description"); return 1; }
  if (out2 && !ws) { set_error("fused 3-D dual pass: residuals need a workspace"); return 1; }
  FusedArgs<T> a = make_fused_args<T>(d);
  const bool vec = vec3_ok<T>(d, y_new, y, xn, xo);
  const int V = vec ? VecOf<T>::N : 1;
  const size_t rb = (d->ny + (size_t)kBlock * V - 1) / ((size_t)kBlock * V);
  a.cols_per_block = pick_cols3d(d->nx, rb, d->L, out2 != nullptr);
  dim3 grid((unsigned)rb, (unsigned)((d->nx + a.cols_per_block - 1) / a.cols_per_block), (unsigned)d->L), block(kBlock);
  if (out2 && (size_t)grid.x * grid.y * grid.z > (size_t)kReduceBlocks) { set_error("fused 3-D dual pass: grid exceeds the reduction workspace"); return 1; }
  hipStream_t s = as_stream(stream);
  double* partial = static_cast<double*>(ws);
  const UniformProx<T> uf = make_uniform_prox<T>(a.f_val, (T)sigma * a.Sval);
  const bool f_uniform = !d->f_coeff_ptr[0] && !d->f_coeff_ptr[2] && !d->f_coeff_ptr[4];
  const bool fast = d->f_fn == PROST_FN_IND_LEQ0;
#define GO(VECv, FFNv, RESv) PH_LAUNCH((fused_dual3d_kernel<T, VECv, FFNv, RESv>), grid, block, 0, s, y_new, y, xn, xo, a, uf, f_uniform, (T)sigma, (T)theta, use_kx_prev != 0, partial)
  if (vec) {
    if (fast) { if (out2) GO(VecOf<T>::N, PROST_FN_IND_LEQ0, true); else GO(VecOf<T>::N, PROST_FN_IND_LEQ0, false); }
    else { if (out2) GO(VecOf<T>::N, -1, true); else GO(VecOf<T>::N, -1, false); }
  } else {
    if (out2) GO(1, -1, true); else GO(1, -1, false);
  }
#undef GO
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "fused 3-D dual pass"); }
  if (out2) return launch_fold(out2, partial, grid.x * grid.y * grid.z, false, s);
  return 0;
}

template int run_primal3d<float>(const prost_hip_fused_desc*, float*, const float*, const float*, const float*, double, int, int, double*, void*, void*);
template int run_primal3d<double>(const prost_hip_fused_desc*, double*, const double*, const double*, const double*, double, int, int, double*, void*, void*);
template int run_dual3d<float>(const prost_hip_fused_desc*, float*, const float*, const float*, const float*, double, double, int, double*, void*, void*);
template int run_dual3d<double>(const prost_hip_fused_desc*, double*, const double*, const double*, const double*, double, double, int, double*, void*, void*);

}  // namespace prost_hip
