// kernels_linop.hip -- generic linear-operator block kernels for gfx950.
//
// Layout facts used throughout (reference block_gradient2d.cu:46-56): an image is stored
// column-major, idx = y + x*ny + l*nx*ny (label_first: idx = l + y*L + x*ny*L), so the
// contiguous axis is y (resp. (y,l)).  Every kernel maps consecutive lanes to consecutive
// addresses of that axis: a wave reads/writes 256 contiguous bytes (fp32) per access, and the
// +-1 / +-ny neighbours of a lane are the same or the adjacent cache lines (served by L1/L2).
#include "common.hpp"

namespace prost_hip {

// ------------------------------------------------------------------------------------------
// gradient 2-D / 3-D, forward and adjoint (block_gradient2d.cu:26-139, block_gradient3d.cu:25-150)
// grid: x = tiles of the contiguous run (ny, or ny*L when label_first), y = image column x, z = l
// ------------------------------------------------------------------------------------------
template <class T, bool LF, bool D3, bool ACC>
__global__ void __launch_bounds__(kBlock) grad_fwd_kernel(T* __restrict__ res, const T* __restrict__ rhs,
                                                          size_t nx, size_t ny, size_t L) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t x = blockIdx.y;
  size_t y, l, idx, sy, sx, sl;
  if (LF) {
    if (t >= ny * L) return;
    l = t % L; y = t / L;
    idx = l + y * L + x * ny * L; sy = L; sx = ny * L; sl = 1;
  } else {
    if (t >= ny) return;
    y = t; l = blockIdx.z;
    idx = y + x * ny + l * nx * ny; sy = 1; sx = ny; sl = nx * ny;
  }
  const size_t N = nx * ny * L;
  const T val = rhs[idx];
  T gx = 0, gy = 0;
  if (y < ny - 1) gy = rhs[idx + sy] - val;
  if (x < nx - 1) gx = rhs[idx + sx] - val;
  if (ACC) { res[idx] += gx; res[idx + N] += gy; } else { res[idx] = gx; res[idx + N] = gy; }
  if (D3) {
    T gl;
    if (l < L - 1) gl = rhs[idx + sl] - val; else gl = -val;     // Dirichlet (block_gradient3d.cu:73-76)
    if (ACC) res[idx + 2 * N] += gl; else res[idx + 2 * N] = gl;
  }
}

template <class T, bool LF, bool D3, bool ACC>
__global__ void __launch_bounds__(kBlock) grad_adj_kernel(T* __restrict__ res, const T* __restrict__ rhs,
                                                          size_t nx, size_t ny, size_t L) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t x = blockIdx.y;
  size_t y, l, idx, sy, sx, sl;
  if (LF) {
    if (t >= ny * L) return;
    l = t % L; y = t / L;
    idx = l + y * L + x * ny * L; sy = L; sx = ny * L; sl = 1;
  } else {
    if (t >= ny) return;
    y = t; l = blockIdx.z;
    idx = y + x * ny + l * nx * ny; sy = 1; sx = ny; sl = nx * ny;
  }
  const size_t N = nx * ny * L;
  T divx, divy;
  if (y < ny - 1) divy = rhs[idx + N]; else divy = 0;
  if (y > 0) divy -= rhs[idx + N - sy];
  if (x < nx - 1) divx = rhs[idx]; else divx = 0;
  if (x > 0) divx -= rhs[idx - sx];
  T s;
  if (D3) {
    T divl = rhs[idx + 2 * N];
    if (l > 0) divl -= rhs[idx + 2 * N - sl];
    s = divx + divy + divl;
  } else {
    s = divx + divy;
  }
  if (ACC) res[idx] -= s; else res[idx] = (T)0 - s;     // adjoint is minus the divergence
}

template <class T, bool D3>
static int launch_grad(bool adjoint, T* res, const T* rhs, size_t nx, size_t ny, size_t L, int lf, int acc, void* stream) {
  if (nx == 0 || ny == 0 || L == 0) return 0;
  if (nx > 65535 || (!lf && L > 65535)) { set_error("gradient: nx and L must be <= 65535"); return 1; }
  dim3 block(kBlock), grid;
  if (lf) grid = dim3((unsigned)((ny * L + kBlock - 1) / kBlock), (unsigned)nx, 1);
  else grid = dim3((unsigned)((ny + kBlock - 1) / kBlock), (unsigned)nx, (unsigned)L);
  hipStream_t s = as_stream(stream);
#define GO(K, LFv, ACCv) hipLaunchKernelGGL((K<T, LFv, D3, ACCv>), grid, block, 0, s, res, rhs, nx, ny, L)
  if (!adjoint) {
    if (lf) { if (acc) GO(grad_fwd_kernel, true, true); else GO(grad_fwd_kernel, true, false); }
    else { if (acc) GO(grad_fwd_kernel, false, true); else GO(grad_fwd_kernel, false, false); }
  } else {
    if (lf) { if (acc) GO(grad_adj_kernel, true, true); else GO(grad_adj_kernel, true, false); }
    else { if (acc) GO(grad_adj_kernel, false, true); else GO(grad_adj_kernel, false, false); }
  }
#undef GO
  PH_LAUNCH_END("gradient kernel");
}

// ------------------------------------------------------------------------------------------
// multi-diagonal operator (block_diags.cu:36-96).  The (offset, factor) band table is staged in
// LDS once per workgroup (<= 1024 entries = 12 KiB) instead of CUDA __constant__ memory.
// ------------------------------------------------------------------------------------------
constexpr int kMaxDiags = 1024;   // block_diags.cu:28

template <class T, bool ADJ>
__global__ void __launch_bounds__(kBlock) diags_kernel(T* __restrict__ res, const T* __restrict__ rhs, size_t nrows,
                                                       size_t ncols, int ndiags, const int64_t* __restrict__ offsets,
                                                       const float* __restrict__ factors, size_t limit) {
  __shared__ int64_t s_ofs[kMaxDiags];
  __shared__ float s_fac[kMaxDiags];
  for (int i = threadIdx.x; i < ndiags; i += kBlock) { s_ofs[i] = offsets[i]; s_fac[i] = factors[i]; }
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < limit; i += (size_t)gridDim.x * kBlock) {
    T result = 0;
    if (!ADJ) {
      const long long row = (long long)i;
      for (int d = 0; d < ndiags; d++) {
        const long long col = row + s_ofs[d];
        if (col < 0) continue;
        if (col >= (long long)ncols) break;
        result += rhs[col] * s_fac[d];
      }
    } else {
      const long long col = (long long)i;
      for (int d = 0; d < ndiags; d++) {
        const long long o = s_ofs[d];
        if (o <= col && (col - o) < (long long)nrows && (col - o) >= 0) result += rhs[col - o] * s_fac[d];
        if (o > col) break;
      }
    }
    res[i] += result;
  }
}

template <class T>
static int launch_diags(bool adj, T* res, const T* rhs, size_t nrows, size_t ncols, size_t ndiags, const int64_t* ofs,
                        const float* fac, int quirk, void* stream) {
  if (ndiags >= (size_t)kMaxDiags) { set_error("Out of constant memory. Too many BlockDiags or too many diagonals."); return 1; }
  size_t limit = adj ? ncols : nrows;
  if (adj && quirk) { size_t g = ((nrows + 255) / 256) * 256; if (g < limit) limit = g; }   // block_diags.cu:210-211
  if (limit == 0) return 0;
  if (adj) hipLaunchKernelGGL((diags_kernel<T, true>), dim3(grid_for(limit)), dim3(kBlock), 0, as_stream(stream), res, rhs, nrows, ncols, (int)ndiags, ofs, fac, limit);
  else hipLaunchKernelGGL((diags_kernel<T, false>), dim3(grid_for(limit)), dim3(kBlock), 0, as_stream(stream), res, rhs, nrows, ncols, (int)ndiags, ofs, fac, limit);
  PH_LAUNCH_END("diags kernel");
}

// ------------------------------------------------------------------------------------------
// CSR SpMV, res += A rhs (cusparse<t>csrmv alpha = beta = 1, block_sparse.cu:156-168).
// LANES lanes cooperate on one row (1 = row per lane, strictly sequential sum; 64 = one
// wavefront per row) chosen from the mean row length; partial sums fold with wave shuffles.
// ------------------------------------------------------------------------------------------
template <class T, int LANES>
__global__ void __launch_bounds__(kBlock) csr_spmv_kernel(T* __restrict__ res, const T* __restrict__ rhs, size_t nrows,
                                                          const T* __restrict__ val, const int32_t* __restrict__ ptr,
                                                          const int32_t* __restrict__ ind) {
  const size_t gtid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (LANES - 1);
  const size_t rows_per_pass = (size_t)gridDim.x * kBlock / LANES;
  for (size_t row = gtid / LANES; row < nrows; row += rows_per_pass) {
    const int32_t b = ptr[row], e = ptr[row + 1];
    T sum = 0;
    for (int32_t j = b + lane; j < e; j += LANES) sum += val[j] * rhs[ind[j]];
    if (LANES > 1) {
#pragma unroll
      for (int o = LANES / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, LANES);
    }
    if (lane == 0) res[row] += sum;
  }
}

template <class T>
static int launch_csr(T* res, const T* rhs, size_t nrows, size_t nnz, const T* val, const int32_t* ptr, const int32_t* ind, void* stream) {
  if (nrows == 0) return 0;
  const double mean = (double)nnz / (double)nrows;
  hipStream_t s = as_stream(stream);
  if (mean <= 6.0) hipLaunchKernelGGL((csr_spmv_kernel<T, 1>), dim3(grid_for(nrows)), dim3(kBlock), 0, s, res, rhs, nrows, val, ptr, ind);
  else if (mean <= 24.0) hipLaunchKernelGGL((csr_spmv_kernel<T, 4>), dim3(grid_for(nrows * 4)), dim3(kBlock), 0, s, res, rhs, nrows, val, ptr, ind);
  else if (mean <= 96.0) hipLaunchKernelGGL((csr_spmv_kernel<T, 16>), dim3(grid_for(nrows * 16)), dim3(kBlock), 0, s, res, rhs, nrows, val, ptr, ind);
  else hipLaunchKernelGGL((csr_spmv_kernel<T, 64>), dim3(grid_for(nrows * 64)), dim3(kBlock), 0, s, res, rhs, nrows, val, ptr, ind);
  PH_LAUNCH_END("csr spmv kernel");
}

// ------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kBlock) scale_kernel(T* __restrict__ x, size_t n, T beta) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) x[i] = beta * x[i];
}
template <class T, bool VIA_FLOAT>
__global__ void __launch_bounds__(kBlock) negate_kernel(T* __restrict__ x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    x[i] = VIA_FLOAT ? (T)(-(float)x[i]) : -x[i];
}

template <class T>
static int launch_scale(T* x, size_t n, double beta, void* stream) {
  if (n == 0) return 0;
  if (beta == 0.0) { PH_CHECK(hipMemsetAsync(x, 0, n * sizeof(T), as_stream(stream))); return 0; }
  hipLaunchKernelGGL((scale_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, as_stream(stream), x, n, (T)beta);
  PH_LAUNCH_END("scale kernel");
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_grad2d_fwd_f32(float* r, const float* x, size_t nx, size_t ny, size_t L, int lf, int acc, void* s) { return launch_grad<float, false>(false, r, x, nx, ny, L, lf, acc, s); }
int prost_hip_grad2d_fwd_f64(double* r, const double* x, size_t nx, size_t ny, size_t L, int lf, int acc, void* s) { return launch_grad<double, false>(false, r, x, nx, ny, L, lf, acc, s); }
int prost_hip_grad2d_adj_f32(float* r, const float* x, size_t nx, size_t ny, size_t L, int lf, int acc, void* s) { return launch_grad<float, false>(true, r, x, nx, ny, L, lf, acc, s); }
int prost_hip_grad2d_adj_f64(double* r, const double* x, size_t nx, size_t ny, size_t L, int lf, int acc, void* s) { return launch_grad<double, false>(true, r, x, nx, ny, L, lf, acc, s); }
int prost_hip_grad3d_fwd_f32(float* r, const float* x, size_t nx, size_t ny, size_t L, int lf, int acc, void* s) { return launch_grad<float, true>(false, r, x, nx, ny, L, lf, acc, s); }
int prost_hip_grad3d_fwd_f64(double* r, const double* x, size_t nx, size_t ny, size_t L, int lf, int acc, void* s) { return launch_grad<double, true>(false, r, x, nx, ny, L, lf, acc, s); }
int prost_hip_grad3d_adj_f32(float* r, const float* x, size_t nx, size_t ny, size_t L, int lf, int acc, void* s) { return launch_grad<float, true>(true, r, x, nx, ny, L, lf, acc, s); }
int prost_hip_grad3d_adj_f64(double* r, const double* x, size_t nx, size_t ny, size_t L, int lf, int acc, void* s) { return launch_grad<double, true>(true, r, x, nx, ny, L, lf, acc, s); }

int prost_hip_diags_fwd_f32(float* r, const float* x, size_t nr, size_t nc, size_t nd, const int64_t* o, const float* f, void* s) { return launch_diags<float>(false, r, x, nr, nc, nd, o, f, 0, s); }
int prost_hip_diags_fwd_f64(double* r, const double* x, size_t nr, size_t nc, size_t nd, const int64_t* o, const float* f, void* s) { return launch_diags<double>(false, r, x, nr, nc, nd, o, f, 0, s); }
int prost_hip_diags_adj_f32(float* r, const float* x, size_t nr, size_t nc, size_t nd, const int64_t* o, const float* f, int q, void* s) { return launch_diags<float>(true, r, x, nr, nc, nd, o, f, q, s); }
int prost_hip_diags_adj_f64(double* r, const double* x, size_t nr, size_t nc, size_t nd, const int64_t* o, const float* f, int q, void* s) { return launch_diags<double>(true, r, x, nr, nc, nd, o, f, q, s); }

int prost_hip_csr_spmv_acc_f32(float* r, const float* x, size_t nrows, size_t nnz, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_csr<float>(r, x, nrows, nnz, v, p, i, s); }
int prost_hip_csr_spmv_acc_f64(double* r, const double* x, size_t nrows, size_t nnz, const double* v, const int32_t* p, const int32_t* i, void* s) { return launch_csr<double>(r, x, nrows, nnz, v, p, i, s); }

int prost_hip_scale_f32(float* x, size_t n, double beta, void* s) { return launch_scale<float>(x, n, beta, s); }
int prost_hip_scale_f64(double* x, size_t n, double beta, void* s) { return launch_scale<double>(x, n, beta, s); }
int prost_hip_negate_f32(float* x, size_t n, void* s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL((negate_kernel<float, false>), dim3(grid_for(n)), dim3(kBlock), 0, as_stream(s), x, n);
  PH_LAUNCH_END("negate kernel");
}
int prost_hip_negate_f64(double* x, size_t n, int via_float, void* s) {
  if (n == 0) return 0;
  if (via_float) hipLaunchKernelGGL((negate_kernel<double, true>), dim3(grid_for(n)), dim3(kBlock), 0, as_stream(s), x, n);
  else hipLaunchKernelGGL((negate_kernel<double, false>), dim3(grid_for(n)), dim3(kBlock), 0, as_stream(s), x, n);
  PH_LAUNCH_END("negate kernel");
}
}  // extern "C"
