// kernels_fused_iter3d_pw.hip -- gradient3d, one kernel per PDHG iteration, PLANES ACROSS THE WAVEFRONTS of a workgroup.
//
// kernels_fused_iter3d.hip lets every wavefront recompute x_new of the plane above its own (2x the primal
// arithmetic, 2x the loads through L2).  Here a workgroup of WT wavefronts owns WT - 1 consecutive planes of one
// (row strip, column chunk): wavefront w marches over plane l0 + w with the single-plane register pipeline of
// kernels_fused_iter_mc.hip and publishes x_new of the column it just produced in LDS; after ONE workgroup barrier
// per column its lower neighbour reads it for the third difference.  The last wavefront is a HELPER: it only runs
// the primal step of plane l0 + WT - 1 (no dual step, no stores), so the recomputation drops from 1 plane per plane
// to 1 per WT - 1 planes.  The old iterate of the plane above (x) and y3 of the plane below are loaded directly:
// the sibling wavefront streams the same lines at the same time on the same CU.
// Same per-element arithmetic as kernels_fused3d.hip / kernels_fused_iter3d.hip -> bit-identical iterates.
// No residual variant (residual iterations: kernels_fused_iter3d.hip's RES instance).
#include "fused_common.hpp"

namespace prost_hip {

template <class T, int VEC, bool GB>
struct ColPw {
  T y1[VEC], y2[VEC], y3[VEC], x[VEC], b[GB ? VEC : 1];   // own plane
  T zx[VEC];                                              // x of the plane above (old iterate, for K x_prev)
  T y3m[VEC];                                             // y3 of the plane below
  T up;                                                   // y2 of the row above the wave's first row (lane 0)
};

template <class T, int VEC, int GFN, int FFN, bool GB, bool FAST, int WT>
__global__ void __launch_bounds__(kWave * WT) fused_iter3d_pw_kernel(T* __restrict__ x_new, T* __restrict__ y_new, const T* __restrict__ x,
                                                                    const T* __restrict__ y, FusedArgs<T> a, T tau, T sigma, T theta,
                                                                    UniformProx<T> ug, UniformProx<T> uf, bool use_kty, bool use_kx_prev, const PdhgRecord<T>* __restrict__ rec) {
  if (rec) {                       // device-resident step sizes (fused_common.hpp: PdhgRecord): wave-uniform scalar loads
    if (rec->stop) return;
    tau = rec->p.tau; sigma = rec->p.sigma; theta = rec->p.theta; ug = rec->p.ug; uf = rec->p.uf;
  }
  constexpr int kRowsPerWave = (kWave - 1) * VEC;
  constexpr int kPix = kWave * VEC;
  constexpr int kPlanes = WT - 1;                      // planes a workgroup produces
  __shared__ T s_xn[2][WT][kPix];
  const size_t nx = a.nx, ny = a.ny, L = a.L;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  const unsigned groups = (unsigned)((L + kPlanes - 1) / kPlanes);
  const unsigned total = gridDim.x, chunks = a.chunks;
  const unsigned xcd = blockIdx.x % 8u, q = blockIdx.x / 8u;            // XCD-aware tile order: plane group fastest
  const unsigned tile = xcd * (total / 8u) + (xcd < total % 8u ? xcd : total % 8u) + q;
  const unsigned grp = tile % groups, rest = tile / groups;
  const unsigned chunk = rest % chunks, strip = rest / chunks;
  const size_t l = (size_t)grp * kPlanes + wv;         // this wavefront's plane (may be >= L: idle, still takes part in the barriers)
  const bool plane_ok = l < L;
  const bool helper = wv == WT - 1;                    // primal step only
  const size_t row0 = (size_t)strip * kRowsPerWave + (size_t)lane * VEC;
  const bool active = plane_ok && row0 < ny;
  const bool owner = active && !helper && lane < kWave - 1;
  const size_t xa = (size_t)chunk * a.cols_per_block;
  const size_t xb = xa + a.cols_per_block < nx ? xa + a.cols_per_block : nx;
  const size_t P = nx * ny, N = P * L, plane = (plane_ok ? l : 0) * P;
  const bool has_above = l + 1 < L;
  const T tauT = tau * a.Tval, sigS = sigma * a.Sval;
  const bool tiny_is_zero = a.f_val[1] >= (T)kTinyIsZeroRadius;     // device_math.hpp: norm2_leq0_fast
  const T* y1 = y + plane; const T* y2 = y + N + plane; const T* y3 = y + 2 * N + plane;
  const T* xp = x + plane;
  const T* bp = GB ? a.g_ptr[1] + plane : nullptr;

  typedef ColPw<T, VEC, GB> Col;
  auto load_col = [&](size_t c, Col& in) {
    const size_t o = c * ny + row0;
    ldv<T, VEC>(y1 + o, in.y1); ldv<T, VEC>(y2 + o, in.y2); ldv<T, VEC>(y3 + o, in.y3); ldv<T, VEC>(xp + o, in.x);
    if (GB) ldv<T, GB ? VEC : 1>(bp + o, in.b);
    in.up = (lane == 0 && row0 > 0) ? y2[o - 1] : (T)0;
    if (has_above && !helper) ldv<T, VEC>(xp + P + o, in.zx);
    if (l > 0) ldv<T, VEC>(y3 - P + o, in.y3m);
  };
  // x_new of this plane at column c (backend_pdhg.cu:317-338 with block_gradient3d.cu:127-149 on a zero-filled result)
  auto primal_col = [&](size_t c, const Col& in, const T (&p1)[VEC], bool have_prev, T (&xn)[VEC]) {
    T up = lane_up(in.y2[VEC - 1]);
    if (lane == 0) up = in.up;
    T parg[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const size_t row = row0 + j;
      T divy = (row < ny - 1) ? in.y2[j] : (T)0;
      if (row > 0) divy -= (j > 0 ? in.y2[j > 0 ? j - 1 : 0] : up);
      T divx = (c < nx - 1) ? in.y1[j] : (T)0;
      if (have_prev) divx -= p1[j];
      T divl = in.y3[j];
      if (l > 0) divl -= in.y3m[j];
      const T kty = use_kty ? (T)0 - (divx + divy + divl) : (T)0;
      const T arg = in.x[j] - tauT * kty;
      if (FAST) {
        parg[j] = arg - (GB ? in.b[GB ? j : 0] : a.g_val[1]);
      } else {
        T cf[7];
#pragma unroll
        for (int k = 0; k < 7; k++) cf[k] = a.g_val[k];
        if (GB) cf[1] = in.b[GB ? j : 0];
        xn[j] = elem_1d_u<T, GFN>(a.g_fn, arg, cf, ug);
      }
    }
    if (FAST) {
      T r[VEC];
      div_to_float_exact_vec<VEC>(parg, ug.sq, r);
#pragma unroll
      for (int j = 0; j < VEC; j++) xn[j] = r[j] + (GB ? in.b[GB ? j : 0] : a.g_val[1]);
    }
  };
  auto publish = [&](int buf, const T (&xn)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; j++) s_xn[buf][wv][j * kWave + lane] = xn[j];      // [j][lane]: conflict-free banks
  };

  Col cur = {}, nxt = {};
  T h1[VEC], xn_c[VEC], xn_n[VEC], xz_c[VEC], xz_n[VEC];     // xz: x_new of the plane above at columns c / c+1 (from LDS)
#pragma unroll
  for (int j = 0; j < VEC; j++) { h1[j] = 0; xn_c[j] = 0; xn_n[j] = 0; xz_c[j] = 0; xz_n[j] = 0; }
  if (active) {
    load_col(xa, cur);
    if (xa > 0) ldv<T, VEC>(y1 + (xa - 1) * ny + row0, h1);
    if (xa + 1 < nx) load_col(xa + 1, nxt);
  }
  primal_col(xa, cur, h1, xa > 0, xn_c);
  if (owner) stv_nt<T, VEC>(x_new + plane + xa * ny + row0, xn_c);
  publish((int)(xa & 1), xn_c);
  __syncthreads();
  if (has_above && !helper) {
#pragma unroll
    for (int j = 0; j < VEC; j++) xz_c[j] = s_xn[xa & 1][wv + 1][j * kWave + lane];
  }

  for (size_t c = xa; c < xb; c++) {
    const bool has_next = c + 1 < nx;
    Col pre;
    const bool has_pre = c + 2 < nx && c + 1 < xb;
    if (active && has_pre) load_col(c + 2, pre);
    const int buf = (int)((c + 1) & 1);
    if (has_next) {
      primal_col(c + 1, nxt, cur.y1, true, xn_n);
      if (owner && c + 1 < xb) stv_nt<T, VEC>(x_new + plane + (c + 1) * ny + row0, xn_n);
      publish(buf, xn_n);
    }
    // one barrier per column; the buffer written now (parity of c+1) was last read before the previous barrier
    __syncthreads();
    if (has_next && has_above && !helper) {
#pragma unroll
      for (int j = 0; j < VEC; j++) xz_n[j] = s_xn[buf][wv + 1][j * kWave + lane];
    }
    // ---- dual step of column c (backend_pdhg.cu:341-370 with block_gradient3d.cu:62-80) ----
    const T bel_n = lane_down(xn_c[0]);
    const T bel_o = lane_down(cur.x[0]);
    if (owner) {
      T out[3][VEC];
      T av[FAST ? 3 : 1][FAST ? VEC : 1], nv[FAST ? VEC : 1];
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const size_t row = row0 + j;
        const T below_n = (j < VEC - 1) ? xn_c[j < VEC - 1 ? j + 1 : 0] : bel_n;
        const T below_o = (j < VEC - 1) ? cur.x[j < VEC - 1 ? j + 1 : 0] : bel_o;
        T kx[3], kp[3], arg[3];
        kx[0] = has_next ? xn_n[j] - xn_c[j] : (T)0;
        kx[1] = (row < ny - 1) ? below_n - xn_c[j] : (T)0;
        kx[2] = has_above ? xz_c[j] - xn_c[j] : -xn_c[j];                       // Dirichlet (:73-76)
        kp[0] = (use_kx_prev && has_next) ? nxt.x[j] - cur.x[j] : (T)0;
        kp[1] = (use_kx_prev && row < ny - 1) ? below_o - cur.x[j] : (T)0;
        kp[2] = use_kx_prev ? (has_above ? cur.zx[j] - cur.x[j] : -cur.x[j]) : (T)0;
        const T yv[3] = {cur.y1[j], cur.y2[j], cur.y3[j]};
        T norm = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          arg[i] = yv[i] + sigS * ((1 + theta) * kx[i] - theta * kp[i]);         // backend_pdhg.cu:54-70
          norm += arg[i] * arg[i];
        }
        if (FAST) {
          nv[FAST ? j : 0] = norm;
#pragma unroll
          for (int i = 0; i < 3; i++) av[FAST ? i : 0][FAST ? j : 0] = arg[i];
        } else if (norm > 0) {
          norm = t_sqrt(norm);
          const T pr = scaled_prox_u<T, FFN>(a.f_fn, norm, a.f_val, uf);
#pragma unroll
          for (int i = 0; i < 3; i++) out[i][j] = pr * arg[i] / norm;
        } else {
#pragma unroll
          for (int i = 0; i < 3; i++) out[i][j] = 0;
        }
      }
      if constexpr (FAST) {
        // out = pr v / ||v||, pr = min(||v|| - b, 0) + b, 0 for ||v|| = 0: device_math.hpp
        norm2_leq0_fast<T, 3, VEC>(nv, av, a.f_val[1], tiny_is_zero, out);
      }
      const size_t o = plane + c * ny + row0;
      stv_nt<T, VEC>(y_new + o, out[0]);
      stv_nt<T, VEC>(y_new + N + o, out[1]);
      stv_nt<T, VEC>(y_new + 2 * N + o, out[2]);
    }
    cur = nxt;
    if (has_pre) nxt = pre;
#pragma unroll
    for (int j = 0; j < VEC; j++) { xn_c[j] = xn_n[j]; xz_c[j] = xz_n[j]; }
  }
}

bool fused3d_desc_ok(const prost_hip_fused_desc* d);

template <class T>
static bool iter3d_pw_ok(const prost_hip_fused_desc* d) {
  if (!fused3d_desc_ok(d)) return false;
  if (d->L < 2) return false;
  if (d->ny % VecOf<T>::N != 0 || d->ny < (size_t)VecOf<T>::N) return false;
  for (int k = 0; k < 7; k++) {
    if (d->f_coeff_ptr[k]) return false;
    if (k != 1 && d->g_coeff_ptr[k]) return false;
  }
  if (d->g_coeff_ptr[1] && !aligned16(d->g_coeff_ptr[1])) return false;
  if (d->res_x1 != 0 && !(d->res_x0 == 0 && d->res_x1 >= d->nx)) return false;
  const size_t strips = (d->ny + (size_t)(kWave - 1) * VecOf<T>::N - 1) / ((size_t)(kWave - 1) * VecOf<T>::N);
  return strips * d->L * ((d->nx + 2) / 3) < (size_t)1 << 31;
}

template <class T>
static int run_iter3d_pw(const prost_hip_fused_desc* d, T* x_new, T* y_new, const T* x, const T* y, double tau, double sigma, double theta,
                         int use_kty, int use_kx_prev, int cols, int waves, void* stream, void* record = nullptr) {
  if (!iter3d_pw_ok<T>(d)) { set_error("fused 3-D iteration (planes across wavefronts): unsupported description"); return 1; }
  if (!aligned16(x_new) || !aligned16(y_new) || !aligned16(x) || !aligned16(y)) { set_error("fused 3-D iteration: vectors must be 16-byte aligned"); return 1; }
  if (x_new == x || y_new == y) { set_error("fused 3-D iteration: outputs must not alias inputs"); return 1; }
  constexpr int V = VecOf<T>::N;
  FusedArgs<T> a = make_fused_args<T>(d);
  // measured 2048^2 x 64 fp32 on one box: 4 wavefronts (3 planes + helper) 2.47 ms, 8 wavefronts 2.61 ms, every wavefront
  // recomputing its upper plane (kernels_fused_iter3d.hip) 2.59 ms: the per-column barrier costs more with 8 participants
  const int wt = waves == 4 || waves == 8 ? waves : 4;
  const size_t planes = (size_t)wt - 1, groups = (d->L + planes - 1) / planes;
  const size_t strips = (d->ny + (size_t)(kWave - 1) * V - 1) / ((size_t)(kWave - 1) * V);
  // 2048^2 x 64 fp32, 4 wavefronts: 6 columns 2.13 ms, 12 columns 2.06 ms, 18 columns 2.07 ms
  size_t c = cols > 0 ? (size_t)cols : 12;
  if (cols <= 0) while (c > 3 && strips * groups * ((d->nx + c - 1) / c) * wt < 8192) c -= 3;
  if (c > d->nx) c = d->nx;
  a.cols_per_block = (unsigned)c;
  a.chunks = (unsigned)((d->nx + c - 1) / c);
  const unsigned grid = (unsigned)(strips * a.chunks * groups);
  const UniformProx<T> ug = make_uniform_prox<T>(a.g_val, (T)tau * a.Tval);
  const UniformProx<T> uf = make_uniform_prox<T>(a.f_val, (T)sigma * a.Sval);
  hipStream_t s = as_stream(stream);
  const bool gb = d->g_coeff_ptr[1] != nullptr;
  const bool gsq = d->g_fn == PROST_FN_SQUARE, fle = d->f_fn == PROST_FN_IND_LEQ0;
  const bool fast = gsq && fle && ug.a_one && ug.den_one && !ug.degenerate && a.g_val[3] == (T)0 && uf.a_one && uf.den_one && a.f_val[3] == (T)0;
#define GO2(G, F, B, FASTv, WTv) PH_LAUNCH((fused_iter3d_pw_kernel<T, V, G, F, B, FASTv, WTv>), dim3(grid), dim3(kWave * WTv), 0, s, x_new, y_new, x, y, a, (T)tau, (T)sigma, (T)theta, ug, uf, use_kty != 0, use_kx_prev != 0, static_cast<const PdhgRecord<T>*>(record))
#define GO(G, F, B, FASTv) do { if (wt == 8) GO2(G, F, B, FASTv, 8); else GO2(G, F, B, FASTv, 4); } while (0)
  if (fast) { if (gb) GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, true, true); else GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, false, true); }
  else { if (gb) GO(-1, -1, true, false); else GO(-1, -1, false, false); }
#undef GO
#undef GO2
  PH_LAUNCH_END("fused 3-D iteration kernel (planes across wavefronts)");
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_fused_iteration3d_pw_supported(const prost_hip_fused_desc* d, int dtype) { return (dtype == 0 ? iter3d_pw_ok<float>(d) : iter3d_pw_ok<double>(d)) ? 1 : 0; }
int prost_hip_fused_iteration3d_pw_f32(const prost_hip_fused_desc* d, float* x_new, float* y_new, const float* x, const float* y, double tau, double sigma,
                                       double theta, int use_kty, int use_kx_prev, int cols, int waves, void* stream) {
  return run_iter3d_pw<float>(d, x_new, y_new, x, y, tau, sigma, theta, use_kty, use_kx_prev, cols, waves, stream);
}
int prost_hip_fused_iteration3d_pw_f64(const prost_hip_fused_desc* d, double* x_new, double* y_new, const double* x, const double* y, double tau, double sigma,
                                       double theta, int use_kty, int use_kx_prev, int cols, int waves, void* stream) {
  return run_iter3d_pw<double>(d, x_new, y_new, x, y, tau, sigma, theta, use_kty, use_kx_prev, cols, waves, stream);
}
int prost_hip_fused_iteration3d_pw_rec_f32(const prost_hip_fused_desc* d, float* x_new, float* y_new, const float* x, const float* y, void* record, int use_kty,
                                           int use_kx_prev, int cols, int waves, void* stream) {
  if (!record) { set_error("fused_iteration3d_pw_rec: no record"); return 1; }
  return run_iter3d_pw<float>(d, x_new, y_new, x, y, 1.0, 1.0, 1.0, use_kty, use_kx_prev, cols, waves, stream, record);
}
int prost_hip_fused_iteration3d_pw_rec_f64(const prost_hip_fused_desc* d, double* x_new, double* y_new, const double* x, const double* y, void* record, int use_kty,
                                           int use_kx_prev, int cols, int waves, void* stream) {
  if (!record) { set_error("fused_iteration3d_pw_rec: no record"); return 1; }
  return run_iter3d_pw<double>(d, x_new, y_new, x, y, 1.0, 1.0, 1.0, use_kty, use_kx_prev, cols, waves, stream, record);
}
}  // extern "C"
