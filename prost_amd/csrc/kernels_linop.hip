// kernels_linop.hip -- generic linear-operator block kernels for gfx950.
//
// Layout facts used throughout (reference block_gradient2d.cu:46-56): an image is stored
// column-major, idx = y + x*ny + l*nx*ny (label_first: idx = l + y*L + x*ny*L), so the
// contiguous axis is y (resp. (y,l)).  Every kernel maps consecutive lanes to consecutive
// addresses of that axis: a wave reads/writes 256 contiguous bytes (fp32) per access, and the
// +-1 / +-ny neighbours of a lane are the same or the adjacent cache lines (served by L1/L2).
#include <cstdlib>
#include "elementwise.hpp"
#include "fused_op.hpp"

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

// ---- planar layout, 16 bytes per lane, column marching ------------------------------------------
// A workgroup owns a strip of kBlock*VEC rows and walks `cols` consecutive image columns of one
// label slice.  A lane holds VEC consecutive rows (float4 / double2): the x+1 column it needs for
// the x-difference is the vector it loads for the next step anyway (kept in registers, so every
// input element is fetched once per workgroup), and the +-1 row neighbour comes from the adjacent
// lane by a wave shuffle -- only lane 63 (lane 0 for the adjoint) reads one extra scalar.
// Arithmetic per element is that of the scalar kernels above, so results are bit-identical.
template <class T, int VEC, bool D3, bool ACC>
__global__ void __launch_bounds__(kBlock) grad_fwd_vec_kernel(T* __restrict__ res, const T* __restrict__ rhs,
                                                              size_t nx, size_t ny, size_t L, unsigned strips, unsigned chunks, int cols) {
  const unsigned strip = blockIdx.x % strips, chunk = (blockIdx.x / strips) % chunks;
  const size_t l = blockIdx.x / (strips * chunks);
  const size_t row0 = ((size_t)strip * kBlock + threadIdx.x) * VEC;
  const bool active = row0 < ny;
  const size_t x0 = (size_t)chunk * cols, x1 = x0 + cols < nx ? x0 + cols : nx;
  const size_t N = nx * ny * L, slice = nx * ny;
  const T* src = rhs + l * slice;
  T cur[VEC], nxt[VEC];
#pragma unroll
  for (int j = 0; j < VEC; j++) cur[j] = 0;
  if (active) ldv<T, VEC>(src + x0 * ny + row0, cur);
  for (size_t x = x0; x < x1; x++) {
    const size_t idx = l * slice + x * ny + row0;
    const bool has_next = x + 1 < nx;
#pragma unroll
    for (int j = 0; j < VEC; j++) nxt[j] = 0;
    if (active && has_next) ldv<T, VEC>(src + (x + 1) * ny + row0, nxt);
    const T below = row_below<T, VEC>(cur, src + x * ny, row0, ny, active);
    T gx[VEC], gy[VEC], gl[VEC], up[VEC];
    if (D3) {
#pragma unroll
      for (int j = 0; j < VEC; j++) up[j] = 0;
      if (active && l + 1 < L) ldv<T, VEC>(rhs + idx + slice, up);
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const T val = cur[j];
      const T dn = (j + 1 < VEC) ? cur[(j + 1) % VEC] : below;
      gy[j] = (row0 + j < ny - 1) ? dn - val : (T)0;
      gx[j] = has_next ? nxt[j] - val : (T)0;
      if (D3) gl[j] = (l + 1 < L) ? up[j] - val : -val;
    }
    if (active) {
      if (ACC) {
        T o[VEC];
        ldv<T, VEC>(res + idx, o);
#pragma unroll
        for (int j = 0; j < VEC; j++) gx[j] = o[j] + gx[j];
        ldv<T, VEC>(res + idx + N, o);
#pragma unroll
        for (int j = 0; j < VEC; j++) gy[j] = o[j] + gy[j];
        if (D3) {
          ldv<T, VEC>(res + idx + 2 * N, o);
#pragma unroll
          for (int j = 0; j < VEC; j++) gl[j] = o[j] + gl[j];
        }
      }
      stv<T, VEC>(res + idx, gx);
      stv<T, VEC>(res + idx + N, gy);
      if (D3) stv<T, VEC>(res + idx + 2 * N, gl);
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) cur[j] = nxt[j];
  }
}

template <class T, int VEC, bool D3, bool ACC>
__global__ void __launch_bounds__(kBlock) grad_adj_vec_kernel(T* __restrict__ res, const T* __restrict__ rhs,
                                                              size_t nx, size_t ny, size_t L, unsigned strips, unsigned chunks, int cols) {
  const unsigned strip = blockIdx.x % strips, chunk = (blockIdx.x / strips) % chunks;
  const size_t l = blockIdx.x / (strips * chunks);
  const size_t row0 = ((size_t)strip * kBlock + threadIdx.x) * VEC;
  const bool active = row0 < ny;
  const size_t x0 = (size_t)chunk * cols, x1 = x0 + cols < nx ? x0 + cols : nx;
  const size_t N = nx * ny * L, slice = nx * ny;
  const T* px_ = rhs + l * slice;            // x-component plane of this slice
  const T* py_ = rhs + N + l * slice;        // y-component
  T prev[VEC], px[VEC], py[VEC], pl[VEC], plm[VEC];
#pragma unroll
  for (int j = 0; j < VEC; j++) prev[j] = 0;
  if (active && x0 > 0) ldv<T, VEC>(px_ + (x0 - 1) * ny + row0, prev);
  for (size_t x = x0; x < x1; x++) {
    const size_t idx = l * slice + x * ny + row0;
#pragma unroll
    for (int j = 0; j < VEC; j++) { px[j] = 0; py[j] = 0; pl[j] = 0; plm[j] = 0; }
    if (active) {
      ldv<T, VEC>(px_ + x * ny + row0, px);
      ldv<T, VEC>(py_ + x * ny + row0, py);
      if (D3) {
        ldv<T, VEC>(rhs + 2 * N + idx, pl);
        if (l > 0) ldv<T, VEC>(rhs + 2 * N + idx - slice, plm);
      }
    }
    const T above = row_above<T, VEC>(py, py_ + x * ny, row0, active);
    T o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      T divx, divy;
      if (row0 + j < ny - 1) divy = py[j]; else divy = 0;
      if (row0 + j > 0) divy -= (j > 0) ? py[(j + VEC - 1) % VEC] : above;
      if (x < nx - 1) divx = px[j]; else divx = 0;
      if (x > 0) divx -= prev[j];
      T s;
      if (D3) {
        T divl = pl[j];
        if (l > 0) divl -= plm[j];
        s = divx + divy + divl;
      } else {
        s = divx + divy;
      }
      o[j] = s;
    }
    if (active) {
      if (ACC) {
        T r[VEC];
        ldv<T, VEC>(res + idx, r);
#pragma unroll
        for (int j = 0; j < VEC; j++) o[j] = r[j] - o[j];
      } else {
#pragma unroll
        for (int j = 0; j < VEC; j++) o[j] = (T)0 - o[j];
      }
      stv<T, VEC>(res + idx, o);
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) prev[j] = px[j];
  }
}

// ---- label-first layout (idx = l + y L + x ny L), 2-D, 16 bytes per lane ------------------------------------------------------
// (round 6: the scalar kernel above reached 0.41-0.44 of the HBM peak on this layout, 4-byte accesses.)  A column x is one contiguous
// run of R = ny L values, t = l + y L; the y-neighbour of t is t + L, the x-neighbour t + R.  A lane owns VEC consecutive t: the
// x-neighbours are the vector it loads for the next column anyway, the y-neighbours one more 16-byte access at a 4-byte aligned address
// (gfx950 serves those; the lanes at the end of the run, whose access would leave it, read element by element).  Arithmetic per element
// is that of the scalar kernel: bit-identical.
template <class T, int VEC>
__device__ __forceinline__ void ldv_at(const T* p, T (&v)[VEC]) {          // 16 bytes at any sizeof(T)-aligned address
  typedef typename UVecOf<T>::type V;
  const V t = *reinterpret_cast<const V*>(p);
#pragma unroll
  for (int j = 0; j < VEC; j++) v[j] = t[j];
}
template <class T, int VEC, bool ACC>
__global__ void __launch_bounds__(kBlock) grad_fwd_lf_vec_kernel(T* __restrict__ res, const T* __restrict__ rhs, size_t nx, size_t ny, size_t L, unsigned strips,
                                                                 int cols) {
  const unsigned strip = blockIdx.x % strips, chunk = blockIdx.x / strips;
  const size_t R = ny * L, N = nx * R;
  const size_t t0 = ((size_t)strip * kBlock + threadIdx.x) * VEC;
  if (t0 >= R) return;
  const size_t x0 = (size_t)chunk * cols, x1 = x0 + cols < nx ? x0 + cols : nx;
  T cur[VEC], nxt[VEC], dn[VEC];
  ldv<T, VEC>(rhs + x0 * R + t0, cur);
  for (size_t x = x0; x < x1; x++) {
    const size_t idx = x * R + t0;
    const bool has_next = x + 1 < nx;
#pragma unroll
    for (int j = 0; j < VEC; j++) { nxt[j] = 0; dn[j] = 0; }
    if (has_next) ldv<T, VEC>(rhs + idx + R, nxt);
    if (t0 + VEC - 1 + L < R) ldv_at<T, VEC>(rhs + idx + L, dn);
    else {
#pragma unroll
      for (int j = 0; j < VEC; j++) if (t0 + j + L < R) dn[j] = rhs[idx + j + L];
    }
    T gx[VEC], gy[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const T val = cur[j];
      gy[j] = (t0 + j + L < R) ? dn[j] - val : (T)0;            // y < ny - 1
      gx[j] = has_next ? nxt[j] - val : (T)0;
    }
    if (ACC) {
      T o[VEC];
      ldv<T, VEC>(res + idx, o);
#pragma unroll
      for (int j = 0; j < VEC; j++) gx[j] = o[j] + gx[j];
      ldv<T, VEC>(res + idx + N, o);
#pragma unroll
      for (int j = 0; j < VEC; j++) gy[j] = o[j] + gy[j];
    }
    stv<T, VEC>(res + idx, gx);
    stv<T, VEC>(res + idx + N, gy);
#pragma unroll
    for (int j = 0; j < VEC; j++) cur[j] = nxt[j];
  }
}
template <class T, int VEC, bool ACC>
__global__ void __launch_bounds__(kBlock) grad_adj_lf_vec_kernel(T* __restrict__ res, const T* __restrict__ rhs, size_t nx, size_t ny, size_t L, unsigned strips,
                                                                 int cols) {
  const unsigned strip = blockIdx.x % strips, chunk = blockIdx.x / strips;
  const size_t R = ny * L, N = nx * R;
  const size_t t0 = ((size_t)strip * kBlock + threadIdx.x) * VEC;
  if (t0 >= R) return;
  const size_t x0 = (size_t)chunk * cols, x1 = x0 + cols < nx ? x0 + cols : nx;
  T prev[VEC], px[VEC], py[VEC], pu[VEC];
#pragma unroll
  for (int j = 0; j < VEC; j++) prev[j] = 0;
  if (x0 > 0) ldv<T, VEC>(rhs + (x0 - 1) * R + t0, prev);
  for (size_t x = x0; x < x1; x++) {
    const size_t idx = x * R + t0;
    ldv<T, VEC>(rhs + idx, px);
    ldv<T, VEC>(rhs + N + idx, py);
#pragma unroll
    for (int j = 0; j < VEC; j++) pu[j] = 0;
    if (t0 >= L) ldv_at<T, VEC>(rhs + N + idx - L, pu);
    else {
#pragma unroll
      for (int j = 0; j < VEC; j++) if (t0 + j >= L) pu[j] = rhs[N + idx + j - L];
    }
    T o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      T divx, divy;
      if (t0 + j + L < R) divy = py[j]; else divy = 0;          // y < ny - 1
      if (t0 + j >= L) divy -= pu[j];                            // y > 0
      if (x < nx - 1) divx = px[j]; else divx = 0;
      if (x > 0) divx -= prev[j];
      o[j] = divx + divy;
    }
    if (ACC) {
      T r[VEC];
      ldv<T, VEC>(res + idx, r);
#pragma unroll
      for (int j = 0; j < VEC; j++) o[j] = r[j] - o[j];
    } else {
#pragma unroll
      for (int j = 0; j < VEC; j++) o[j] = (T)0 - o[j];
    }
    stv<T, VEC>(res + idx, o);
#pragma unroll
    for (int j = 0; j < VEC; j++) prev[j] = px[j];
  }
}

// columns per workgroup: non-power-of-two chunks keep concurrently running workgroups on
// different HBM channels (see kernels_fused_iter.hip); shrink until the grid fills 256 CUs x 4
static int pick_grad_cols(size_t nx, size_t strips, size_t L) {
  for (int c : {12, 6, 3}) if (strips * ((nx + c - 1) / c) * L >= 2048) return c;
  return 1;
}

template <class T, bool D3>
static int launch_grad(bool adjoint, T* res, const T* rhs, size_t nx, size_t ny, size_t L, int lf, int acc, void* stream) {
  if (nx == 0 || ny == 0 || L == 0) return 0;
  constexpr int V = VecOf<T>::N;
  if (!lf && ny % V == 0 && aligned16(res) && aligned16(rhs)) {
    const size_t strips = (ny + (size_t)kBlock * V - 1) / ((size_t)kBlock * V);
    const int cols = pick_grad_cols(nx, strips, L);
    const size_t chunks = (nx + cols - 1) / cols, blocks = strips * chunks * L;
    if (blocks < (1ull << 31)) {
      hipStream_t s = as_stream(stream);
#define GOV(K, ACCv) hipLaunchKernelGGL((K<T, V, D3, ACCv>), dim3((unsigned)blocks), dim3(kBlock), 0, s, res, rhs, nx, ny, L, (unsigned)strips, (unsigned)chunks, cols)
      if (!adjoint) { if (acc) GOV(grad_fwd_vec_kernel, true); else GOV(grad_fwd_vec_kernel, false); }
      else { if (acc) GOV(grad_adj_vec_kernel, true); else GOV(grad_adj_vec_kernel, false); }
#undef GOV
      PH_LAUNCH_END("gradient kernel");
    }
  }
  if (lf && !D3 && (ny * L) % V == 0 && aligned16(res) && aligned16(rhs)) {
    const size_t R = ny * L, strips = (R + (size_t)kBlock * V - 1) / ((size_t)kBlock * V);
    const int cols = pick_grad_cols(nx, strips, 1);
    const size_t blocks = strips * ((nx + cols - 1) / cols);
    if (blocks < (1ull << 31)) {
      hipStream_t s = as_stream(stream);
#define GOL(K, ACCv) hipLaunchKernelGGL((K<T, V, ACCv>), dim3((unsigned)blocks), dim3(kBlock), 0, s, res, rhs, nx, ny, L, (unsigned)strips, cols)
      if (!adjoint) { if (acc) GOL(grad_fwd_lf_vec_kernel, true); else GOL(grad_fwd_lf_vec_kernel, false); }
      else { if (acc) GOL(grad_adj_lf_vec_kernel, true); else GOL(grad_adj_lf_vec_kernel, false); }
#undef GOL
      PH_LAUNCH_END("gradient kernel");
    }
  }
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

// 16 bytes of consecutive rows per lane.  A workgroup pass covers kBlock*VEC rows; when every
// diagonal stays inside the operand for the whole pass (all but the first/last few passes) the
// bounds tests and the 64-bit index arithmetic per (row, diagonal) disappear and each diagonal is
// one (4-byte aligned) 16-byte load per lane.  Border passes run the scalar formula above.
template <class T>
struct __attribute__((packed, aligned(sizeof(T)))) UnalignedVec {
  T v[VecOf<T>::N];
};

template <class T, bool ADJ>
__global__ void __launch_bounds__(kBlock) diags_vec_kernel(T* __restrict__ res, const T* __restrict__ rhs, size_t nrows,
                                                           size_t ncols, int ndiags, const int64_t* __restrict__ offsets,
                                                           const float* __restrict__ factors, size_t limit) {
  constexpr int VEC = VecOf<T>::N;
  __shared__ int64_t s_ofs[kMaxDiags];
  __shared__ float s_fac[kMaxDiags];
  __shared__ int64_t s_range[2];
  for (int i = threadIdx.x; i < ndiags; i += kBlock) { s_ofs[i] = offsets[i]; s_fac[i] = factors[i]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t lo = 0, hi = 0;
    for (int d = 0; d < ndiags; d++) { const int64_t o = s_ofs[d]; if (d == 0 || o < lo) lo = o; if (d == 0 || o > hi) hi = o; }
    s_range[0] = lo; s_range[1] = hi;
  }
  __syncthreads();
  const long long omin = s_range[0], omax = s_range[1];
  const size_t span = (size_t)kBlock * VEC;
  for (size_t base = (size_t)blockIdx.x * span; base < limit; base += (size_t)gridDim.x * span) {
    const long long lo = (long long)base, hi = (long long)(base + span - 1);
    const bool interior = hi < (long long)limit &&
                          (ADJ ? (lo - omax >= 0 && hi - omin < (long long)nrows) : (lo + omin >= 0 && hi + omax < (long long)ncols));
    const size_t i0 = base + (size_t)threadIdx.x * VEC;
    if (interior) {
      T r[VEC];
#pragma unroll
      for (int j = 0; j < VEC; j++) r[j] = 0;
      for (int d = 0; d < ndiags; d++) {
        const long long o = s_ofs[d];
        const float f = s_fac[d];
        const UnalignedVec<T> u = *reinterpret_cast<const UnalignedVec<T>*>(rhs + (ADJ ? (long long)i0 - o : (long long)i0 + o));
#pragma unroll
        for (int j = 0; j < VEC; j++) r[j] += u.v[j] * f;
      }
      T o[VEC];
      ldv<T, VEC>(res + i0, o);
#pragma unroll
      for (int j = 0; j < VEC; j++) o[j] += r[j];
      stv<T, VEC>(res + i0, o);
    } else {
      for (int j = 0; j < VEC; j++) {
        const size_t i = i0 + j;
        if (i >= limit) break;
        T result = 0;
        if (!ADJ) {
          for (int d = 0; d < ndiags; d++) {
            const long long col = (long long)i + s_ofs[d];
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
  }
}

template <class T>
static int launch_diags(bool adj, T* res, const T* rhs, size_t nrows, size_t ncols, size_t ndiags, const int64_t* ofs,
                        const float* fac, int quirk, void* stream) {
  if (ndiags >= (size_t)kMaxDiags) { set_error("Out of constant memory. Too many BlockDiags or too many diagonals."); return 1; }
  size_t limit = adj ? ncols : nrows;
  if (adj && quirk) { size_t g = ((nrows + 255) / 256) * 256; if (g < limit) limit = g; }   // block_diags.cu:210-211
  if (limit == 0) return 0;
  constexpr int V = VecOf<T>::N;
  if (aligned16(res) && ndiags > 0) {
    unsigned g = grid_for((limit + V - 1) / V);
    if (ndiags > 16 && g > 8192) g = 8192;     // every workgroup stages the band table: amortise a long one
    if (adj) hipLaunchKernelGGL((diags_vec_kernel<T, true>), dim3(g), dim3(kBlock), 0, as_stream(stream), res, rhs, nrows, ncols, (int)ndiags, ofs, fac, limit);
    else hipLaunchKernelGGL((diags_vec_kernel<T, false>), dim3(g), dim3(kBlock), 0, as_stream(stream), res, rhs, nrows, ncols, (int)ndiags, ofs, fac, limit);
    PH_LAUNCH_END("diags kernel");
  }
  if (adj) hipLaunchKernelGGL((diags_kernel<T, true>), dim3(grid_for(limit)), dim3(kBlock), 0, as_stream(stream), res, rhs, nrows, ncols, (int)ndiags, ofs, fac, limit);
  else hipLaunchKernelGGL((diags_kernel<T, false>), dim3(grid_for(limit)), dim3(kBlock), 0, as_stream(stream), res, rhs, nrows, ncols, (int)ndiags, ofs, fac, limit);
  PH_LAUNCH_END("diags kernel");
}

// ------------------------------------------------------------------------------------------
// Pattern-compressed SpMV (round 3).  Many of the sparse matrices problem descriptions hand to block.sparse are STENCILS written
// out row by row -- spmat_gradient2d / 3d (example_rof_primal.m, example_nonconvex_rof.m), blur operators (example_deblurring.m):
// the rows repeat a handful of (column - row, value) sequences.  Such a matrix is stored as one 16-bit pattern number per row plus
// a small table; a product then streams 2 bytes per row instead of 8 per entry + 4 per row (gradient2d as a matrix: 14 instead of
// 52 bytes per pixel forward, 13 instead of 48 for the stored transpose).  A row is summed in the order of its CSR entries
// (sequentially, the order of the oracle's restatement of csrmv), so the result equals that of csr_spmv_kernel<T, 1, .> bit for
// bit.
// ------------------------------------------------------------------------------------------
// four consecutive elements at an address that is only element-aligned (gfx950 serves such 16-byte accesses): (row + rel) is
// whatever the stencil says
template <class T> struct Quad;
template <> struct Quad<float> { typedef float V __attribute__((ext_vector_type(4), aligned(4))); };
template <> struct Quad<double> { typedef double V __attribute__((ext_vector_type(4), aligned(8))); };

// TAB: the pattern table (at most kTabPatterns patterns, kTabEntries entries) is staged in LDS by the workgroup while the pattern
// numbers of its rows are on their way, and read from there: the chain of dependent memory round trips of a wavefront is numbers ->
// operands -> store instead of numbers -> table offsets -> table entries -> operands -> store.  That chain IS the run time of the
// small products of the generic path (256^2 .. 1024^2: one or two waves per SIMD, ~10 us per launch of which ~4.5 us are the launch
// itself); the arithmetic and its order are unchanged.
constexpr int kTabPatterns = 256, kTabEntries = 1024;
__device__ __forceinline__ float uniform_value(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uniform_value(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <class T, bool ACC, bool TAB>
__global__ void __launch_bounds__(kBlock) pattern_spmv_kernel(T* __restrict__ res, const T* __restrict__ rhs, size_t nrows, const uint16_t* __restrict__ ids,
                                                              const int32_t* __restrict__ pptr_g, const int32_t* __restrict__ rel_g, const T* __restrict__ pval_g,
                                                              int npatterns, int nentries) {
  __shared__ int32_t s_pptr[TAB ? kTabPatterns + 1 : 1];
  __shared__ int32_t s_rel[TAB ? kTabEntries : 1];
  __shared__ T s_val[TAB ? kTabEntries : 1];
  if (TAB) {
    for (int i = threadIdx.x; i <= npatterns; i += kBlock) s_pptr[TAB ? i : 0] = pptr_g[i];
    for (int i = threadIdx.x; i < nentries; i += kBlock) { s_rel[TAB ? i : 0] = rel_g[i]; s_val[TAB ? i : 0] = pval_g[i]; }
  }
  const int32_t* pptr = TAB ? s_pptr : pptr_g;
  const int32_t* rel = TAB ? s_rel : rel_g;
  const T* pval = TAB ? s_val : pval_g;
  // A lane takes 4 consecutive rows, a wavefront G groups of 256 rows.  Nearly always all of them have ONE pattern: its table
  // entries are then wave-uniform (scalar loads) and an entry costs one 16 / 32-byte load of rhs per lane and group.  Mixed
  // wavefronts (the seams of the stencil, the last rows) walk the table per row.
  constexpr int R = 4, G = 2;
  typedef typename Quad<T>::V QV;
  const int lane = threadIdx.x & (kWave - 1);
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kWave, nwaves = (size_t)gridDim.x * kBlock / kWave;
  constexpr size_t kRowsPerWave = (size_t)kWave * R * G;
  unsigned id[G][R];
  auto load_ids = [&](size_t base) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      const size_t row0 = base + ((size_t)g * kWave + lane) * R;
      if (row0 + R <= nrows) {
        const uint2 w = *reinterpret_cast<const uint2*>(ids + row0);        // row0 is a multiple of 4 and ids 8-byte aligned
        id[g][0] = w.x & 0xFFFFu; id[g][1] = w.x >> 16; id[g][2] = w.y & 0xFFFFu; id[g][3] = w.y >> 16;
      } else {
#pragma unroll
        for (int j = 0; j < R; j++) id[g][j] = row0 + j < nrows ? (unsigned)ids[row0 + j] : 0xFFFFFFFFu;   // past the end: no pattern, no uniform wavefront
      }
    }
  };
  size_t base = wave * kRowsPerWave;
  if (base < nrows) load_ids(base);                     // requested before the table is waited for
  if (TAB) __syncthreads();                             // the table is in LDS (every wavefront of the workgroup arrives here)
  while (base < nrows) {
    bool same = true;
    const unsigned id0 = (unsigned)__builtin_amdgcn_readfirstlane((int)id[0][0]);
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int j = 0; j < R; j++) same = same && id[g][j] == id0;
    T out[G][R];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int j = 0; j < R; j++) out[g][j] = 0;
    if (__builtin_amdgcn_ballot_w64(!same) == 0) {
      // (wave-uniform table entries: scalar loads from global memory, broadcast reads + readfirstlane from LDS)
      const int32_t b = TAB ? __builtin_amdgcn_readfirstlane(pptr[id0]) : pptr[id0], e = TAB ? __builtin_amdgcn_readfirstlane(pptr[id0 + 1]) : pptr[id0 + 1];
      for (int32_t k = b; k < e; k++) {
        const long r = (long)(TAB ? __builtin_amdgcn_readfirstlane(rel[k]) : rel[k]);
        const T v = TAB ? uniform_value(pval[k]) : pval[k];
#pragma unroll
        for (int g = 0; g < G; g++) {
          const QV x = *reinterpret_cast<const QV*>(rhs + (long)(base + ((size_t)g * kWave + lane) * R) + r);
#pragma unroll
          for (int j = 0; j < R; j++) out[g][j] += v * x[j];
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int j = 0; j < R; j++) {
          if (id[g][j] == 0xFFFFFFFFu) continue;
          const long row = (long)(base + ((size_t)g * kWave + lane) * R + j);
          const int32_t b = pptr[id[g][j]], e = pptr[id[g][j] + 1];
          T sum = 0;
          for (int32_t k = b; k < e; k++) sum += pval[k] * rhs[row + rel[k]];
          out[g][j] = sum;
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
      const size_t row0 = base + ((size_t)g * kWave + lane) * R;
      if (row0 + R <= nrows) {
        QV o;
        if (ACC) o = *reinterpret_cast<const QV*>(res + row0);
#pragma unroll
        for (int j = 0; j < R; j++) o[j] = (ACC ? o[j] : (T)0) + out[g][j];
        *reinterpret_cast<QV*>(res + row0) = o;
      } else {
#pragma unroll
        for (int j = 0; j < R; j++) if (row0 + j < nrows) res[row0 + j] = (ACC ? res[row0 + j] : (T)0) + out[g][j];
      }
    }
    base += nwaves * kRowsPerWave;
    if (base < nrows) load_ids(base);
  }
}
// Round 5: the same product with the row walk of the fused kernels (fused_op.hpp: pattern_rows) -- a lane takes VEC consecutive rows, the
// wavefront one pass per distinct pattern among its rows, the table read through the constant address space with scalar loads, the
// operands of up to six entries in flight together.  Same sums in the same order as pattern_spmv_kernel.
// TAB: the table (<= kTabPatterns patterns, <= kTabEntries entries) is staged in LDS by the workgroup while the first pattern numbers are on
// their way: a pass then reads its offsets and entries as LDS broadcasts instead of two dependent scalar round trips
template <class T, bool ACC, int VEC, bool TAB>
__global__ void __launch_bounds__(kBlock) pattern_spmv_rows_kernel(T* __restrict__ res, const T* __restrict__ rhs, size_t nrows, const uint16_t* __restrict__ ids,
                                                                    const int32_t* __restrict__ pptr, const int32_t* __restrict__ rel, const T* __restrict__ pval,
                                                                    const int32_t* __restrict__ anchor, int npatterns, int nentries) {
  constexpr int V = VEC;
  __shared__ int32_t s_pptr[TAB ? kTabPatterns + 1 : 1];
  __shared__ int32_t s_rel[TAB ? kTabEntries : 1];
  __shared__ T s_val[TAB ? kTabEntries : 1];
  if (TAB) {
    for (int i = threadIdx.x; i <= npatterns; i += kBlock) s_pptr[TAB ? i : 0] = pptr[i];
    for (int i = threadIdx.x; i < nentries; i += kBlock) { s_rel[TAB ? i : 0] = rel[i]; s_val[TAB ? i : 0] = pval[i]; }
    __syncthreads();
  }
  const size_t nvec = nrows / V;
  const T* rr[1] = {rhs};
  for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < nvec; g += (size_t)gridDim.x * kBlock) {
    const size_t r0 = g * V;
    T sum[1][V], o[V];
    if (ACC) ldv<T, V>(res + r0, o);
    if (TAB) pattern_rows<T, V, 1>(ids, (const int32_t*)s_pptr, (const int32_t*)s_rel, (const T*)s_val, rr, r0, sum, anchor);
    else
    pattern_rows<T, V, 1>(ids, as_constant(pptr), as_constant(rel), as_constant(pval), rr, r0, sum, anchor);
#pragma unroll
    for (int j = 0; j < V; j++) o[j] = (ACC ? o[j] : (T)0) + sum[0][j];
    stv<T, V>(res + r0, o);
  }
  if (blockIdx.x == 0 && threadIdx.x < nrows - nvec * V) {              // the last nrows % V rows
    const size_t r0 = nvec * V + threadIdx.x;
    T sum[1][1];
    if (TAB) pattern_rows<T, 1, 1>(ids, (const int32_t*)s_pptr, (const int32_t*)s_rel, (const T*)s_val, rr, r0, sum, anchor);
    else pattern_rows<T, 1, 1>(ids, as_constant(pptr), as_constant(rel), as_constant(pval), rr, r0, sum, anchor);
    res[r0] = (ACC ? res[r0] : (T)0) + sum[0][0];
  }
}

template <class T>
static int launch_pattern(T* res, const T* rhs, size_t nrows, const uint16_t* ids, const int32_t* pptr, const int32_t* rel, const T* pval, int acc, void* stream,
                          int npatterns = 0, int nentries = 0, const int32_t* anchor = nullptr) {
  if (nrows == 0) return 0;
  if (!res || !rhs || !ids || !pptr || !rel || !pval) { set_error("pattern spmv: null pointer"); return 1; }
  if (reinterpret_cast<uintptr_t>(ids) % 8 != 0) { set_error("pattern spmv: the pattern numbers must be 8-byte aligned"); return 1; }
  hipStream_t s = as_stream(stream);
  // round 5: 16 bytes of rows per lane, one pass per distinct pattern of a wavefront, scalar table loads (pattern_spmv_rows_kernel); same
  // box, deblurring's shape with separate products: 256^2 17 992 -> 25 093, 1024^2 11 061 -> 14 344, 2048^2 4 557 -> 5 177 iterations/s.
  // (Requesting the operands of the table's most frequent pattern before the pattern numbers arrive was tried on top of this and in
  // the prox kernels that apply the operator: 3-7 % SLOWER at every size -- with scalar table loads the walk is cheap, the speculation is not)
  if (anchor && reinterpret_cast<uintptr_t>(anchor) % 16 != 0) { set_error("pattern spmv: the anchors must be 16-byte aligned"); return 1; }
  if (reinterpret_cast<uintptr_t>(res) % 16 == 0) {
    const size_t lanes = nrows / VecOf<T>::N;
    size_t gx = (lanes + kBlock - 1) / kBlock;
    if (gx < 1) gx = 1;
    if (gx > 16384) gx = 16384;
    // (the table in LDS pays for small products only: same box, deblurring's shape, 256^2 24.4 k -> 25.1 k iterations/s, 1024^2 14.2 k -> 14.0 k,
    // 2048^2 5.54 k -> 5.21 k -- beyond a few hundred thousand rows the scalar walk, whose table sits in the scalar cache, is the faster one)
    const bool tab2 = npatterns > 0 && npatterns <= kTabPatterns && nentries > 0 && nentries <= kTabEntries && nrows <= ((size_t)1 << 18);
    if (tab2) {
      if (acc) hipLaunchKernelGGL((pattern_spmv_rows_kernel<T, true, VecOf<T>::N, true>), dim3((unsigned)gx), dim3(kBlock), 0, s, res, rhs, nrows, ids, pptr, rel, pval, anchor, npatterns, nentries);
      else hipLaunchKernelGGL((pattern_spmv_rows_kernel<T, false, VecOf<T>::N, true>), dim3((unsigned)gx), dim3(kBlock), 0, s, res, rhs, nrows, ids, pptr, rel, pval, anchor, npatterns, nentries);
    } else {
      if (acc) hipLaunchKernelGGL((pattern_spmv_rows_kernel<T, true, VecOf<T>::N, false>), dim3((unsigned)gx), dim3(kBlock), 0, s, res, rhs, nrows, ids, pptr, rel, pval, anchor, 0, 0);
      else hipLaunchKernelGGL((pattern_spmv_rows_kernel<T, false, VecOf<T>::N, false>), dim3((unsigned)gx), dim3(kBlock), 0, s, res, rhs, nrows, ids, pptr, rel, pval, anchor, 0, 0);
    }
    PH_LAUNCH_END("pattern spmv rows kernel");
  }
  if (anchor) {                  // (res not 16-byte aligned: the same walk one row per lane)
    size_t gx = (nrows + kBlock - 1) / kBlock;
    if (gx > 65536) gx = 65536;
    if (acc) hipLaunchKernelGGL((pattern_spmv_rows_kernel<T, true, 1, false>), dim3((unsigned)gx), dim3(kBlock), 0, s, res, rhs, nrows, ids, pptr, rel, pval, anchor, 0, 0);
    else hipLaunchKernelGGL((pattern_spmv_rows_kernel<T, false, 1, false>), dim3((unsigned)gx), dim3(kBlock), 0, s, res, rhs, nrows, ids, pptr, rel, pval, anchor, 0, 0);
    PH_LAUNCH_END("pattern spmv rows kernel");
  }
  const unsigned grid = grid_for((nrows + 7) / 8);
  // (res not 16-byte aligned: the row-by-row kernel) the table staged in LDS where its size is known and fits (prost_hip_pattern_spmv_tab)
  const bool tab = npatterns > 0 && npatterns <= kTabPatterns && nentries > 0 && nentries <= kTabEntries && nrows <= ((size_t)1 << 22);
  if (tab) {
    if (acc) hipLaunchKernelGGL((pattern_spmv_kernel<T, true, true>), dim3(grid), dim3(kBlock), 0, s, res, rhs, nrows, ids, pptr, rel, pval, npatterns, nentries);
    else hipLaunchKernelGGL((pattern_spmv_kernel<T, false, true>), dim3(grid), dim3(kBlock), 0, s, res, rhs, nrows, ids, pptr, rel, pval, npatterns, nentries);
  }
  else if (acc) hipLaunchKernelGGL((pattern_spmv_kernel<T, true, false>), dim3(grid), dim3(kBlock), 0, s, res, rhs, nrows, ids, pptr, rel, pval, 0, 0);
  else hipLaunchKernelGGL((pattern_spmv_kernel<T, false, false>), dim3(grid), dim3(kBlock), 0, s, res, rhs, nrows, ids, pptr, rel, pval, 0, 0);
  PH_LAUNCH_END("pattern spmv kernel");
}

// ------------------------------------------------------------------------------------------
// CSR SpMV, res += A rhs (cusparse<t>csrmv alpha = beta = 1, block_sparse.cu:156-168).
// LANES lanes cooperate on one row (1 = row per lane, strictly sequential sum; 64 = one
// wavefront per row) chosen from the mean row length; partial sums fold with wave shuffles.
// ------------------------------------------------------------------------------------------
template <class T, int LANES, bool ACC>
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
    if (lane == 0) res[row] = (ACC ? res[row] : (T)0) + sum;       // ACC = false: the zero fill + accumulate of Block::EvalLocal in one pass
  }
}

template <class T, bool ACC>
static int launch_csr(T* res, const T* rhs, size_t nrows, size_t nnz, const T* val, const int32_t* ptr, const int32_t* ind, void* stream) {
  if (nrows == 0) return 0;
  const double mean = (double)nnz / (double)nrows;      // (example_deblurring.m's 35-tap rows, same box: 1 / 4 / 16 / 64 lanes per row = 593 / 3 969 / 4 853 / 2 319 iterations/s)
  hipStream_t s = as_stream(stream);
  if (mean <= 6.0) hipLaunchKernelGGL((csr_spmv_kernel<T, 1, ACC>), dim3(grid_for(nrows)), dim3(kBlock), 0, s, res, rhs, nrows, val, ptr, ind);
  else if (mean <= 24.0) hipLaunchKernelGGL((csr_spmv_kernel<T, 4, ACC>), dim3(grid_for(nrows * 4)), dim3(kBlock), 0, s, res, rhs, nrows, val, ptr, ind);
  else if (mean <= 96.0) hipLaunchKernelGGL((csr_spmv_kernel<T, 16, ACC>), dim3(grid_for(nrows * 16)), dim3(kBlock), 0, s, res, rhs, nrows, val, ptr, ind);
  else hipLaunchKernelGGL((csr_spmv_kernel<T, 64, ACC>), dim3(grid_for(nrows * 64)), dim3(kBlock), 0, s, res, rhs, nrows, val, ptr, ind);
  PH_LAUNCH_END("csr spmv kernel");
}

// ------------------------------------------------------------------------------------------
// Kronecker blocks: kron(K, I_d) (BlockSparseKronIdKernel, block_sparse_kron_id.cu:26-49) and
// kron(I_d, K) (BlockIdKronSparseKernel, block_id_kron_sparse.cu:26-52); K in CSR with FLOAT values
// whatever T is.  One output element per lane.  kron(K, I): the lanes of a wave share the CSR row
// (ptr / ind / val are wave-uniform broadcasts) and read d-contiguous rhs entries -> fully coalesced.
// kron(I, K): consecutive lanes own consecutive rows of one K copy (a gather, as in the reference).
// ------------------------------------------------------------------------------------------
template <class T, bool ID_FIRST, bool ACC>
__global__ void __launch_bounds__(kBlock) kron_spmv_kernel(T* __restrict__ res, const T* __restrict__ rhs, size_t diaglength, size_t nrows,
                                                           size_t ncols, const float* __restrict__ val, const int32_t* __restrict__ ptr,
                                                           const int32_t* __restrict__ ind) {
  const size_t total = diaglength * nrows;
  // (K's arrays are never written by a kernel: read through the constant address space -- kron(K, I): the row is the same for the lanes of a
  // wavefront wherever diaglength >= 64, its entries then come through scalar loads)
  const PROST_CONSTANT float* cval = as_constant(val);
  const PROST_CONSTANT int32_t* cptr = as_constant(ptr);
  const PROST_CONSTANT int32_t* cind = as_constant(ind);
  for (size_t tx = (size_t)blockIdx.x * kBlock + threadIdx.x; tx < total; tx += (size_t)gridDim.x * kBlock) {
    size_t row, col_ofs;
    if (ID_FIRST) { row = tx % nrows; col_ofs = (tx / nrows) * ncols; }
    else { col_ofs = tx % diaglength; row = tx / diaglength; }
    T sum = 0;
    const int32_t stop = cptr[row + 1];
    for (int32_t i = cptr[row]; i < stop; i++) sum += cval[i] * rhs[ID_FIRST ? (size_t)cind[i] + col_ofs : (size_t)cind[i] * diaglength + col_ofs];
    res[tx] = (ACC ? res[tx] : (T)0) + sum;       // ACC = false: the zero fill + accumulate of Block::EvalLocal in one pass
  }
}

// kron(K, I_d), 16 bytes per lane (round 6; the kernel above: 0.18 of the HBM peak on compulsory bytes at K 12 x 16, d = 2^20 -- a 64-bit
// division per element, 4-byte accesses, and every operand segment re-read from memory once per row that refers to it).  A lane owns V
// consecutive offsets o of the identity and walks the rows of a row chunk (blockIdx.y): K's row is wave-uniform (scalar loads), the operand
// segments of one row chunk are re-read by the SAME lanes shortly after one another (cache hits).  Per output the products are added in CSR
// order into a sum that starts at zero, as above: bit-identical.
template <class T, bool ACC>
__global__ void __launch_bounds__(kBlock) kron_id_vec_kernel(T* __restrict__ res, const T* __restrict__ rhs, unsigned dvec, size_t diaglength,
                                                             unsigned nrows, unsigned rows_per_chunk, const float* __restrict__ val,
                                                             const int32_t* __restrict__ ptr, const int32_t* __restrict__ ind) {
  constexpr int V = 16 / sizeof(T);
  typedef T TV __attribute__((ext_vector_type(V)));
  const PROST_CONSTANT float* cval = as_constant(val);
  const PROST_CONSTANT int32_t* cptr = as_constant(ptr);
  const PROST_CONSTANT int32_t* cind = as_constant(ind);
  const unsigned r0 = blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk < nrows ? r0 + rows_per_chunk : nrows;
  for (unsigned ov = blockIdx.x * kBlock + threadIdx.x; ov < dvec; ov += gridDim.x * kBlock) {
    const size_t o = (size_t)ov * V;
    int32_t i = cptr[r0];
    for (unsigned r = r0; r < r1; r++) {
      const int32_t stop = cptr[r + 1];
      T sum[V];
#pragma unroll
      for (int j = 0; j < V; j++) sum[j] = 0;
      for (; i < stop; i++) {
        const T v = (T)cval[i];
        const TV xv = *reinterpret_cast<const TV*>(rhs + (size_t)cind[i] * diaglength + o);
#pragma unroll
        for (int j = 0; j < V; j++) sum[j] += v * xv[j];
      }
      TV* out = reinterpret_cast<TV*>(res + (size_t)r * diaglength + o);
      TV t;
      if (ACC) t = *out;
#pragma unroll
      for (int j = 0; j < V; j++) t[j] = (ACC ? t[j] : (T)0) + sum[j];
      *out = t;
    }
  }
}

// kron(I_d, K) through LDS (round 6; the kernel above: 0.15 of the HBM peak at K 12 x 16 -- per-lane chains ptr -> ind / val -> operand through
// memory, a 64-bit division per element, 4-byte gathers).  A workgroup takes kCopies consecutive copies of K at a time: their operand range
// [j0 ncols, (j0 + kCopies) ncols) and their output range are CONTIGUOUS, so the operands enter LDS with 16-byte loads and every lane
// writes V consecutive outputs; K itself (ptr, ind, val) is staged in LDS once per workgroup when it fits (kKronMaxNnz / kKronMaxRows), else
// read through the constant address space.  Same products in the same order per output: bit-identical.
constexpr int kKronTile = 4096;        // operand elements per tile (16 KB fp32, 32 KB fp64)
constexpr int kKronMaxNnz = 2048;
constexpr int kKronMaxRows = 1024;
template <class T, bool ACC>
__global__ void __launch_bounds__(kBlock) id_kron_lds_kernel(T* __restrict__ res, const T* __restrict__ rhs, size_t diaglength, unsigned nrows,
                                                             unsigned ncols, unsigned copies, const float* __restrict__ val,
                                                             const int32_t* __restrict__ ptr, const int32_t* __restrict__ ind) {
  constexpr int V = 16 / sizeof(T);
  typedef T TV __attribute__((ext_vector_type(V)));
  __shared__ __attribute__((aligned(16))) T s_rhs[kKronTile];
  __shared__ float s_val[kKronMaxNnz];
  __shared__ int32_t s_ind[kKronMaxNnz];
  __shared__ int32_t s_ptr[kKronMaxRows + 1];
  const PROST_CONSTANT float* cval = as_constant(val);
  const PROST_CONSTANT int32_t* cptr = as_constant(ptr);
  const PROST_CONSTANT int32_t* cind = as_constant(ind);
  const int32_t nnz = cptr[nrows];
  const bool staged = nnz <= kKronMaxNnz;                                       // workgroup-uniform
  for (unsigned k = threadIdx.x; k <= nrows; k += kBlock) s_ptr[k] = cptr[k];
  if (staged)
    for (int32_t k = threadIdx.x; k < nnz; k += kBlock) { s_val[k] = cval[k]; s_ind[k] = cind[k]; }
  // the V outputs of a lane within a tile: t = V (threadIdx.x + kBlock k) + e  ->  copy t / nrows, row t % nrows, advanced without divisions
  const unsigned step_q = (V * kBlock) / nrows, step_m = (V * kBlock) % nrows;
  const unsigned q0 = (V * threadIdx.x) / nrows, m0 = (V * threadIdx.x) % nrows;
  const size_t tiles = (diaglength + copies - 1) / copies;
  // the operands of a tile travel global -> registers -> LDS, and the loads of the NEXT tile are issued before the products of the current
  // one, so they are in flight during them (loaded, then computed with nothing in flight: 0.34 of the HBM peak at S 12 x 16)
  constexpr int NP = kKronTile / (V * kBlock);
  TV pre[NP];
  auto fetch = [&](size_t tl) {
    const size_t j0 = tl * copies;
    const unsigned nc = diaglength - j0 < copies ? (unsigned)(diaglength - j0) : copies;
    const unsigned n_in = nc * ncols;
    const T* __restrict__ src = rhs + j0 * ncols;
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const unsigned k = V * (threadIdx.x + p * kBlock);
      if (k + V <= n_in) pre[p] = *reinterpret_cast<const TV*>(src + k);
      else {
#pragma unroll
        for (int e = 0; e < V; e++) pre[p][e] = k + e < n_in ? src[k + e] : (T)0;
      }
    }
  };
  if (blockIdx.x < tiles) fetch(blockIdx.x);
  for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const size_t j0 = tile * copies;
    const unsigned nc = diaglength - j0 < copies ? (unsigned)(diaglength - j0) : copies;
    const unsigned n_out = nc * nrows;
    T* __restrict__ dst = res + j0 * nrows;
    __syncthreads();                                                            // the previous tile's reads of s_rhs (and the staging of K)
#pragma unroll
    for (int p = 0; p < NP; p++) *reinterpret_cast<TV*>(s_rhs + V * (threadIdx.x + p * kBlock)) = pre[p];
    __syncthreads();
    if (tile + gridDim.x < tiles) fetch(tile + gridDim.x);
    unsigned q = q0, m = m0;
    for (unsigned t = V * threadIdx.x; t < n_out; t += V * kBlock) {
      T sum[V];
      unsigned qe = q, me = m;
#pragma unroll
      for (int e = 0; e < V; e++) {
        sum[e] = 0;
        if (t + e < n_out) {
          const T* __restrict__ xs = s_rhs + qe * ncols;
          if (staged) { for (int32_t i = s_ptr[me], stop = s_ptr[me + 1]; i < stop; i++) sum[e] += (T)s_val[i] * xs[s_ind[i]]; }
          else { for (int32_t i = s_ptr[me], stop = s_ptr[me + 1]; i < stop; i++) sum[e] += (T)cval[i] * xs[cind[i]]; }
        }
        if (++me == nrows) { me = 0; qe++; }
      }
      if (t + V <= n_out) {
        TV* out = reinterpret_cast<TV*>(dst + t);
        TV o;
        if (ACC) o = *out;
#pragma unroll
        for (int e = 0; e < V; e++) o[e] = (ACC ? o[e] : (T)0) + sum[e];
        *out = o;
      } else {
        for (unsigned e = 0; t + e < n_out; e++) dst[t + e] = (ACC ? dst[t + e] : (T)0) + sum[e];
      }
      q += step_q; m += step_m;
      if (m >= nrows) { m -= nrows; q++; }
    }
  }
}

template <class T>
static int launch_kron(bool id_first, T* res, const T* rhs, size_t diaglength, size_t nrows, size_t ncols, const float* val, const int32_t* ptr,
                       const int32_t* ind, void* stream, bool acc = true) {
  const size_t total = diaglength * nrows;
  if (total == 0) return 0;
  constexpr size_t V = 16 / sizeof(T);
  const bool aligned = ((reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(rhs)) & 15u) == 0;
  static const bool plain_only = getenv("PROST_KRON_PLAIN") != nullptr;       // A/B switch (tools/bench_kernels.py)
  if (!plain_only && !id_first && aligned && diaglength % V == 0 && diaglength / V < (1u << 31) && nrows < (1u << 31)) {
    const unsigned dvec = (unsigned)(diaglength / V);
    const unsigned gx = grid_for(dvec);
    // enough workgroups to fill the chip when the identity is short: the rows are cut into chunks (blockIdx.y)
    unsigned gy = gx >= 2048u ? 1u : (2048u + gx - 1) / gx;
    if (gy > nrows) gy = (unsigned)nrows;
    if (gy > 65535u) gy = 65535u;
    const unsigned rpc = (unsigned)((nrows + gy - 1) / gy);
    gy = (unsigned)((nrows + rpc - 1) / rpc);
    if (acc) hipLaunchKernelGGL((kron_id_vec_kernel<T, true>), dim3(gx, gy), dim3(kBlock), 0, as_stream(stream), res, rhs, dvec, diaglength, (unsigned)nrows, rpc, val, ptr, ind);
    else hipLaunchKernelGGL((kron_id_vec_kernel<T, false>), dim3(gx, gy), dim3(kBlock), 0, as_stream(stream), res, rhs, dvec, diaglength, (unsigned)nrows, rpc, val, ptr, ind);
    PH_LAUNCH_END("kronecker spmv kernel (identity last, 16 bytes per lane)");
  }
  const size_t widest = nrows > ncols ? nrows : ncols;
  if (!plain_only && id_first && aligned && nrows >= 1 && ncols >= 1 && widest <= (size_t)kKronMaxRows) {
    // copies per tile: a multiple of 4 (tile starts stay 16-byte aligned whatever nrows / ncols are) that fills the operand tile
    const unsigned copies = (unsigned)((size_t)kKronTile / widest) & ~3u;
    const size_t tiles = (diaglength + copies - 1) / copies;
    const unsigned grid = (unsigned)(tiles < 1024 ? tiles : 1024);       // resident workgroups (4 per CU) that walk the tiles, the next one's loads in flight
    if (acc) hipLaunchKernelGGL((id_kron_lds_kernel<T, true>), dim3(grid), dim3(kBlock), 0, as_stream(stream), res, rhs, diaglength, (unsigned)nrows, (unsigned)ncols, copies, val, ptr, ind);
    else hipLaunchKernelGGL((id_kron_lds_kernel<T, false>), dim3(grid), dim3(kBlock), 0, as_stream(stream), res, rhs, diaglength, (unsigned)nrows, (unsigned)ncols, copies, val, ptr, ind);
    PH_LAUNCH_END("kronecker spmv kernel (identity first, LDS tiles)");
  }
  if (id_first && acc) hipLaunchKernelGGL((kron_spmv_kernel<T, true, true>), dim3(grid_for(total)), dim3(kBlock), 0, as_stream(stream), res, rhs, diaglength, nrows, ncols, val, ptr, ind);
  else if (id_first) hipLaunchKernelGGL((kron_spmv_kernel<T, true, false>), dim3(grid_for(total)), dim3(kBlock), 0, as_stream(stream), res, rhs, diaglength, nrows, ncols, val, ptr, ind);
  else if (acc) hipLaunchKernelGGL((kron_spmv_kernel<T, false, true>), dim3(grid_for(total)), dim3(kBlock), 0, as_stream(stream), res, rhs, diaglength, nrows, ncols, val, ptr, ind);
  else hipLaunchKernelGGL((kron_spmv_kernel<T, false, false>), dim3(grid_for(total)), dim3(kBlock), 0, as_stream(stream), res, rhs, diaglength, nrows, ncols, val, ptr, ind);
  PH_LAUNCH_END("kronecker spmv kernel");
}

// ------------------------------------------------------------------------------------------
template <class T> struct ScaleF { T beta; __device__ T operator()(const T* a) const { return beta * a[0]; } };
template <class T, bool VIA_FLOAT> struct NegateF { __device__ T operator()(const T* a) const { return VIA_FLOAT ? (T)(-(float)a[0]) : -a[0]; } };

template <class T>
static int launch_scale(T* x, size_t n, double beta, void* stream) {
  if (n == 0) return 0;
  if (beta == 0.0) { PH_CHECK(hipMemsetAsync(x, 0, n * sizeof(T), as_stream(stream))); return 0; }
  return launch_ew<T, 1>("scale kernel", x, EwIn<T, 1>{{x}}, n, ScaleF<T>{(T)beta}, as_stream(stream));
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

int prost_hip_csr_spmv_acc_f32(float* r, const float* x, size_t nrows, size_t nnz, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_csr<float, true>(r, x, nrows, nnz, v, p, i, s); }
int prost_hip_csr_spmv_acc_f64(double* r, const double* x, size_t nrows, size_t nnz, const double* v, const int32_t* p, const int32_t* i, void* s) { return launch_csr<double, true>(r, x, nrows, nnz, v, p, i, s); }
int prost_hip_csr_spmv_f32(float* r, const float* x, size_t nrows, size_t nnz, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_csr<float, false>(r, x, nrows, nnz, v, p, i, s); }
int prost_hip_csr_spmv_f64(double* r, const double* x, size_t nrows, size_t nnz, const double* v, const int32_t* p, const int32_t* i, void* s) { return launch_csr<double, false>(r, x, nrows, nnz, v, p, i, s); }
int prost_hip_pattern_spmv_f32(float* r, const float* x, size_t nrows, const uint16_t* ids, const int32_t* pptr, const int32_t* rel, const float* pval, int acc, void* s) { return launch_pattern<float>(r, x, nrows, ids, pptr, rel, pval, acc, s); }
int prost_hip_pattern_spmv_f64(double* r, const double* x, size_t nrows, const uint16_t* ids, const int32_t* pptr, const int32_t* rel, const double* pval, int acc, void* s) { return launch_pattern<double>(r, x, nrows, ids, pptr, rel, pval, acc, s); }
int prost_hip_pattern_spmv_tab_f32(float* r, const float* x, size_t nrows, const uint16_t* ids, const int32_t* pptr, const int32_t* rel, const float* pval, int npatterns, int nentries, int acc, void* s) { return launch_pattern<float>(r, x, nrows, ids, pptr, rel, pval, acc, s, npatterns, nentries); }
int prost_hip_pattern_spmv_tab_f64(double* r, const double* x, size_t nrows, const uint16_t* ids, const int32_t* pptr, const int32_t* rel, const double* pval, int npatterns, int nentries, int acc, void* s) { return launch_pattern<double>(r, x, nrows, ids, pptr, rel, pval, acc, s, npatterns, nentries); }
int prost_hip_pattern_spmv_anchored_f32(float* r, const float* x, size_t nrows, const uint16_t* ids, const int32_t* anchor, const int32_t* pptr, const int32_t* rel, const float* pval, int npatterns,
                                        int nentries, int acc, void* s) {
  if (!anchor) { set_error("pattern spmv: anchors required"); return 1; }
  return launch_pattern<float>(r, x, nrows, ids, pptr, rel, pval, acc, s, npatterns, nentries, anchor);
}
int prost_hip_pattern_spmv_anchored_f64(double* r, const double* x, size_t nrows, const uint16_t* ids, const int32_t* anchor, const int32_t* pptr, const int32_t* rel, const double* pval, int npatterns,
                                        int nentries, int acc, void* s) {
  if (!anchor) { set_error("pattern spmv: anchors required"); return 1; }
  return launch_pattern<double>(r, x, nrows, ids, pptr, rel, pval, acc, s, npatterns, nentries, anchor);
}

int prost_hip_sparse_kron_id_acc_f32(float* r, const float* x, size_t d, size_t nrows, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_kron<float>(false, r, x, d, nrows, 0, v, p, i, s); }
int prost_hip_sparse_kron_id_acc_f64(double* r, const double* x, size_t d, size_t nrows, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_kron<double>(false, r, x, d, nrows, 0, v, p, i, s); }
int prost_hip_id_kron_sparse_acc_f32(float* r, const float* x, size_t d, size_t nrows, size_t ncols, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_kron<float>(true, r, x, d, nrows, ncols, v, p, i, s); }
int prost_hip_id_kron_sparse_acc_f64(double* r, const double* x, size_t d, size_t nrows, size_t ncols, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_kron<double>(true, r, x, d, nrows, ncols, v, p, i, s); }
int prost_hip_sparse_kron_id_f32(float* r, const float* x, size_t d, size_t nrows, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_kron<float>(false, r, x, d, nrows, 0, v, p, i, s, false); }
int prost_hip_sparse_kron_id_f64(double* r, const double* x, size_t d, size_t nrows, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_kron<double>(false, r, x, d, nrows, 0, v, p, i, s, false); }
int prost_hip_id_kron_sparse_f32(float* r, const float* x, size_t d, size_t nrows, size_t ncols, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_kron<float>(true, r, x, d, nrows, ncols, v, p, i, s, false); }
int prost_hip_id_kron_sparse_f64(double* r, const double* x, size_t d, size_t nrows, size_t ncols, const float* v, const int32_t* p, const int32_t* i, void* s) { return launch_kron<double>(true, r, x, d, nrows, ncols, v, p, i, s, false); }

int prost_hip_scale_f32(float* x, size_t n, double beta, void* s) { return launch_scale<float>(x, n, beta, s); }
int prost_hip_scale_f64(double* x, size_t n, double beta, void* s) { return launch_scale<double>(x, n, beta, s); }
int prost_hip_negate_f32(float* x, size_t n, void* s) {
  return launch_ew<float, 1>("negate kernel", x, EwIn<float, 1>{{x}}, n, NegateF<float, false>{}, as_stream(s));
}
int prost_hip_negate_f64(double* x, size_t n, int via_float, void* s) {
  if (via_float) return launch_ew<double, 1>("negate kernel", x, EwIn<double, 1>{{x}}, n, NegateF<double, true>{}, as_stream(s));
  return launch_ew<double, 1>("negate kernel", x, EwIn<double, 1>{{x}}, n, NegateF<double, false>{}, as_stream(s));
}
}  // extern "C"
