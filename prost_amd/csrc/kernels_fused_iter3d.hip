// kernels_fused_iter3d.hip -- ONE kernel per PDHG iteration for gradient3d problems (volumetric TV, BASELINE config 3).
//
// The two-pass scheme (kernels_fused3d.hip) moves 14 floats/voxel/iteration because x_new makes a round trip
// through HBM between the primal and the dual pass.  Here a wavefront owns a strip of rows of ONE plane l and
// marches over a chunk of columns exactly like kernels_fused_iter.hip (x_new of column c+1 one step ahead of
// y_new of column c, row neighbours by wave shuffle, lane 63 a halo lane, one halo column per chunk).  The third
// difference couples plane l with plane l+1:  y_new(l) needs x_new(l+1), which belongs to another wavefront.
// It is RECOMPUTED here from plane l+1's inputs (y1, y2, y3, x, f of l+1; y3 of l is already in registers):
// twice the primal arithmetic (the path is HBM-bound, the VALU has room), and the loads of plane l+1 are the ones
// the wavefront of plane l+1 issues for itself at about the same time -- tiles are ordered plane-after-plane
// inside a (row strip, column chunk) and handed to one XCD, so they meet in that XCD's L2 (or the Infinity
// Cache).  HBM sees
//     read y1, y2, y3, x, f of plane l once;  write x_new, y1, y2, y3   = 9 floats / voxel / iteration (vs 14).
// Per-element arithmetic is that of kernels_fused3d.hip (block_gradient3d.cu:57-80, :127-149 inlined; prox
// expressions of device_math.hpp): the iterates are bit-identical to the two-pass path and to the oracle.
// Residual iterations (RES) additionally stream y_prev of the wavefront's own plane and accumulate the four residual sums.
#include "fused_common.hpp"
#include "reduce.hpp"

namespace prost_hip {

template <class T, int VEC, bool GB, bool RES>
struct Col3 {
  T y1[VEC], y2[VEC], y3[VEC], x[VEC], b[GB ? VEC : 1];        // plane l
  T zy1[VEC], zy2[VEC], zy3[VEC], zx[VEC], zb[GB ? VEC : 1];   // plane l+1 (zero when l is the last plane)
  T y3m[VEC];                                                  // y3 of plane l-1
  T up, zup;                                                   // y2 of the row above the wave's first row (lane 0)
  T p1[RES ? VEC : 1], p2[RES ? VEC : 1], p3[RES ? VEC : 1], p3m[RES ? VEC : 1], pup;   // y_prev of plane l (and y_prev3 of l-1): RES only
};

// GB: the coefficient b of prox_g (the data term f of ROF) is a per-voxel vector; a, c, d, e, alpha, beta of
// prox_g and all coefficients of prox_f* are scalars (host-checked), so the uniform prox forms apply.
// FAST: straight-line instance for the ROF shape (host-checked: prox_g square with scalar a = 1, d = e = 0, prox_f*
// ind_leq0 with scalar a = 1, d = e = 0): the correctly rounded short forms of device_math.hpp instead of the IEEE
// division / sqrt expansions (3 divisions + 1 sqrt per voxel and one fp64 division per primal step otherwise) --
// the generic instance is VALU-bound at ~250 instructions per voxel.
// RES = true (residual iterations) additionally streams y_prev of plane l and accumulates the four residual sums of
// backend_pdhg.cu:392-431 for the wavefront's own plane (one partial of 4 doubles per wavefront, folded by fold4).
template <class T, int VEC, int GFN, int FFN, bool GB, bool FAST, bool RES>
__global__ void __launch_bounds__(kWave) fused_iter3d_kernel(T* __restrict__ x_new, T* __restrict__ y_new, const T* __restrict__ x,
                                                             const T* __restrict__ y, const T* __restrict__ y_prev, FusedArgs<T> a, T tau, T sigma,
                                                             T theta, UniformProx<T> ug, UniformProx<T> uf, bool use_kty, bool use_kx_prev,
                                                             bool use_kty_prev, double* __restrict__ partial, const PdhgRecord<T>* __restrict__ rec) {
  if (rec) {                       // device-resident step sizes (fused_common.hpp: PdhgRecord): wave-uniform scalar loads
    if (rec->stop) return;
    tau = rec->p.tau; sigma = rec->p.sigma; theta = rec->p.theta; ug = rec->p.ug; uf = rec->p.uf;
  }
  const size_t nx = a.nx, ny = a.ny, L = a.L;
  const int lane = threadIdx.x;
  constexpr int kRowsPerWave = (kWave - 1) * VEC;
  // tile order: plane fastest, then column chunk, then row strip; each XCD (workgroup b runs on XCD b % 8) gets a
  // contiguous range of tiles, so the wavefronts of planes l and l+1 of one (strip, chunk) share an L2
  const unsigned total = gridDim.x, chunks = a.chunks;
  const unsigned xcd = blockIdx.x % 8u, q = blockIdx.x / 8u;
  const unsigned tile = xcd * (total / 8u) + (xcd < total % 8u ? xcd : total % 8u) + q;
  const size_t l = tile % L;
  const unsigned rest = tile / (unsigned)L;
  const unsigned chunk = rest % chunks, strip = rest / chunks;
  const size_t row0 = (size_t)strip * kRowsPerWave + (size_t)lane * VEC;
  const bool active = row0 < ny;                       // loads + primal steps
  const bool owner = active && lane < kWave - 1;       // stores
  const size_t xa = (size_t)chunk * a.cols_per_block;
  const size_t xb = xa + a.cols_per_block < nx ? xa + a.cols_per_block : nx;
  const size_t P = nx * ny, N = P * L, plane = l * P;
  const bool has_above = l + 1 < L;
  const T tauT = tau * a.Tval, sigS = sigma * a.Sval;
  const T* y1 = y + plane; const T* y2 = y + N + plane; const T* y3 = y + 2 * N + plane;
  const T* xp = x + plane;
  const T* bp = GB ? a.g_ptr[1] + plane : nullptr;
  const T* q1 = RES ? y_prev + plane : nullptr; const T* q2 = RES ? y_prev + N + plane : nullptr; const T* q3 = RES ? y_prev + 2 * N + plane : nullptr;
  const T sqT = t_sqrt(a.Tval), sqS = t_sqrt(a.Sval);
  const bool tiny_is_zero = a.f_val[1] >= (T)kTinyIsZeroRadius;     // device_math.hpp: norm2_leq0_fast
  const SharedDivisor<T> div_tauT(tau * sqT), div_sigS(sigma * sqS);        // wave-uniform residual divisors: exact quotients through one double reciprocal each
  double r_pd = 0, r_pv = 0, r_dd = 0, r_dv = 0;       // primal diff^2, primal var^2, dual diff^2, dual var^2

  typedef Col3<T, VEC, GB, RES> Col;
  auto load_col = [&](size_t c, Col& in) {
    const size_t o = c * ny + row0;
    ldv<T, VEC>(y1 + o, in.y1); ldv<T, VEC>(y2 + o, in.y2); ldv<T, VEC>(y3 + o, in.y3); ldv<T, VEC>(xp + o, in.x);
    if (GB) ldv<T, GB ? VEC : 1>(bp + o, in.b);
    in.up = (lane == 0 && row0 > 0) ? y2[o - 1] : (T)0;
    if (has_above) {
      ldv<T, VEC>(y1 + P + o, in.zy1); ldv<T, VEC>(y2 + P + o, in.zy2); ldv<T, VEC>(y3 + P + o, in.zy3); ldv<T, VEC>(xp + P + o, in.zx);
      if (GB) ldv<T, GB ? VEC : 1>(bp + P + o, in.zb);
      in.zup = (lane == 0 && row0 > 0) ? y2[P + o - 1] : (T)0;
    }
    if (l > 0) ldv<T, VEC>(y3 - P + o, in.y3m);
    if (RES) {
      ldv<T, RES ? VEC : 1>(q1 + o, in.p1); ldv<T, RES ? VEC : 1>(q2 + o, in.p2); ldv<T, RES ? VEC : 1>(q3 + o, in.p3);
      if (l > 0) ldv<T, RES ? VEC : 1>(q3 - P + o, in.p3m);
      in.pup = (lane == 0 && row0 > 0) ? q2[o - 1] : (T)0;
    }
  };
  // x_new of one plane at column c (backend_pdhg.cu:317-338 with block_gradient3d.cu:127-149 on a zero-filled result)
  auto primal_col = [&](size_t c, const T (&v1)[VEC], const T (&v2)[VEC], const T (&v3)[VEC], const T (&v3m)[VEC], bool has_below_plane,
                        const T (&p1)[VEC], bool have_prev, T upv, const T (&xv)[VEC], const T (&bv)[GB ? VEC : 1], T (&xn)[VEC],
                        T (&ktyv)[RES ? VEC : 1]) {
    T up = lane_up(v2[VEC - 1]);
    if (lane == 0) up = upv;
    T parg[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const size_t row = row0 + j;
      T divy = (row < ny - 1) ? v2[j] : (T)0;
      if (row > 0) divy -= (j > 0 ? v2[j > 0 ? j - 1 : 0] : up);
      T divx = (c < nx - 1) ? v1[j] : (T)0;
      if (have_prev) divx -= p1[j];
      T divl = v3[j];
      if (has_below_plane) divl -= v3m[j];
      const T kty = use_kty ? (T)0 - (divx + divy + divl) : (T)0;
      if (RES) ktyv[RES ? j : 0] = kty;
      const T arg = xv[j] - tauT * kty;
      if (FAST) {
        parg[j] = arg - (GB ? bv[GB ? j : 0] : a.g_val[1]);
      } else {
        T cf[7];
#pragma unroll
        for (int k = 0; k < 7; k++) cf[k] = a.g_val[k];
        if (GB) cf[1] = bv[GB ? j : 0];
        xn[j] = elem_1d_u<T, GFN>(a.g_fn, arg, cf, ug);
      }
    }
    if (FAST) {        // a (v - d tau) = v, fp64 denominator 1: F_prox(v - b; step) + b with the exact reciprocal-based quotient
      T r[VEC];
      div_to_float_exact_vec<VEC>(parg, ug.sq, r);
#pragma unroll
      for (int j = 0; j < VEC; j++) xn[j] = r[j] + (GB ? bv[GB ? j : 0] : a.g_val[1]);
    }
  };

  // dual_residual_transform (backend_pdhg.cu:73-94) for the own plane at column c
  auto dual_residual = [&](size_t c, const Col& in, const T (&pp1)[RES ? VEC : 1], bool have_prev, const T (&xn)[VEC], const T (&ktyv)[RES ? VEC : 1], bool counted) {
    T upp = lane_up(in.p2[RES ? VEC - 1 : 0]);
    if (lane == 0) upp = in.pup;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const size_t row = row0 + j;
      const int jj = RES ? j : 0;
      T dpy = (row < ny - 1) ? in.p2[jj] : (T)0;
      if (row > 0) dpy -= (j > 0 ? in.p2[RES && j > 0 ? j - 1 : 0] : upp);
      T dpx = (c < nx - 1) ? in.p1[jj] : (T)0;
      if (have_prev) dpx -= pp1[jj];
      T dpl = in.p3[jj];
      if (l > 0) dpl -= in.p3m[jj];
      const T ktyp = use_kty_prev ? (T)0 - (dpx + dpy + dpl) : (T)0;
      const T w_hat = div_tauT.div(in.x[j] - xn[j]) - sqT * ktyp;
      const T diff = w_hat + sqT * ktyv[jj];
      if (owner && counted) { r_dd += (double)(diff * diff); r_dv += (double)(w_hat * w_hat); }
    }
  };

  Col cur = {}, nxt = {};
  T hp1[RES ? VEC : 1];                                  // y_prev1 of column xa-1
  T kt_c[RES ? VEC : 1], kt_n[RES ? VEC : 1], kt_z[RES ? VEC : 1];
  T h1[VEC], hz1[VEC];                                  // y1 of column xa-1, planes l and l+1
  T xn_c[VEC], xn_n[VEC], xz_c[VEC], xz_n[VEC];          // x_new of plane l / l+1 at columns c / c+1
#pragma unroll
  for (int j = 0; j < VEC; j++) { h1[j] = 0; hz1[j] = 0; xn_c[j] = 0; xn_n[j] = 0; xz_c[j] = 0; xz_n[j] = 0; }
  if (active) {
    load_col(xa, cur);
    if (xa > 0) {
      ldv<T, VEC>(y1 + (xa - 1) * ny + row0, h1);
      if (has_above) ldv<T, VEC>(y1 + P + (xa - 1) * ny + row0, hz1);
      if (RES) ldv<T, RES ? VEC : 1>(q1 + (xa - 1) * ny + row0, hp1);
    }
    if (xa + 1 < nx) load_col(xa + 1, nxt);
  }
#pragma unroll
  for (int j = 0; j < (RES ? VEC : 1); j++) { if (!(active && xa > 0)) hp1[j] = 0; kt_c[j] = 0; kt_n[j] = 0; kt_z[j] = 0; }
  primal_col(xa, cur.y1, cur.y2, cur.y3, cur.y3m, l > 0, h1, xa > 0, cur.up, cur.x, cur.b, xn_c, kt_c);      // shuffles inside: every lane takes part
  if (RES) dual_residual(xa, cur, hp1, xa > 0, xn_c, kt_c, true);
  if (has_above) primal_col(xa, cur.zy1, cur.zy2, cur.zy3, cur.y3, true, hz1, xa > 0, cur.zup, cur.zx, cur.zb, xz_c, kt_z);
  if (owner) stv_nt<T, VEC>(x_new + plane + xa * ny + row0, xn_c);

  for (size_t c = xa; c < xb; c++) {
    const bool has_next = c + 1 < nx;
    Col pre;                                               // prefetch column c+2 while columns c+1 / c are processed
    const bool has_pre = c + 2 < nx && c + 1 < xb;
    if (active && has_pre) load_col(c + 2, pre);
    if (has_next) {
      primal_col(c + 1, nxt.y1, nxt.y2, nxt.y3, nxt.y3m, l > 0, cur.y1, true, nxt.up, nxt.x, nxt.b, xn_n, kt_n);
      if (RES) dual_residual(c + 1, nxt, cur.p1, true, xn_n, kt_n, c + 1 < xb);
      if (has_above && c + 1 < xb) primal_col(c + 1, nxt.zy1, nxt.zy2, nxt.zy3, nxt.y3, true, cur.zy1, true, nxt.zup, nxt.zx, nxt.zb, xz_n, kt_z);
      if (owner && c + 1 < xb) stv_nt<T, VEC>(x_new + plane + (c + 1) * ny + row0, xn_n);
    }
    // ---- dual step of column c (backend_pdhg.cu:341-370 with block_gradient3d.cu:62-80) ----
    const T bel_n = lane_down(xn_c[0]);
    const T bel_o = lane_down(cur.x[0]);
    if (owner) {
      T out[3][VEC];
      T av[FAST ? 3 : 1][FAST ? VEC : 1], nv[FAST ? VEC : 1];
      T kxv[RES ? 3 : 1][RES ? VEC : 1], kpv[RES ? 3 : 1][RES ? VEC : 1];
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
          if (RES) { kxv[RES ? i : 0][RES ? j : 0] = kx[i]; kpv[RES ? i : 0][RES ? j : 0] = kp[i]; }
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
      if (RES) {                                            // primal_residual_transform (backend_pdhg.cu:97-120)
#pragma unroll
        for (int j = 0; j < VEC; j++) {
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const T yo = i == 0 ? cur.y1[j] : i == 1 ? cur.y2[j] : cur.y3[j];
            const T kxi = kxv[RES ? i : 0][RES ? j : 0], kpi = kpv[RES ? i : 0][RES ? j : 0];
            const T z_hat = div_sigS.div(yo - out[i][j]) + sqS * ((1 + theta) * kxi - theta * kpi);
            const T diff = z_hat - sqS * kxi;
            r_pd += (double)(diff * diff); r_pv += (double)(z_hat * z_hat);
          }
        }
      }
      const size_t o = plane + c * ny + row0;
      stv_nt<T, VEC>(y_new + o, out[0]);
      stv_nt<T, VEC>(y_new + N + o, out[1]);
      stv_nt<T, VEC>(y_new + 2 * N + o, out[2]);
    }
    // shift the pipeline
    cur = nxt;
    if (has_pre) nxt = pre;
#pragma unroll
    for (int j = 0; j < VEC; j++) { xn_c[j] = xn_n[j]; xz_c[j] = xz_n[j]; }
  }
  if (RES) {
    r_pd = wave_sum(r_pd); r_pv = wave_sum(r_pv); r_dd = wave_sum(r_dd); r_dv = wave_sum(r_dv);
    if (lane == 0) {
      double* p = partial + 4 * (size_t)blockIdx.x;
      p[0] = r_pd; p[1] = r_pv; p[2] = r_dd; p[3] = r_dv;
    }
  }
}

bool fused3d_desc_ok(const prost_hip_fused_desc* d);

template <class T>
static bool iter3d_ok(const prost_hip_fused_desc* d) {
  if (!fused3d_desc_ok(d)) return false;
  if (d->ny % VecOf<T>::N != 0 || d->ny < (size_t)VecOf<T>::N) return false;
  for (int k = 0; k < 7; k++) {
    if (d->f_coeff_ptr[k]) return false;                       // prox_f*: scalar coefficients only
    if (k != 1 && d->g_coeff_ptr[k]) return false;             // prox_g: only b may be a per-voxel vector
  }
  if (d->g_coeff_ptr[1] && !aligned16(d->g_coeff_ptr[1])) return false;
  if (d->res_x1 != 0 && !(d->res_x0 == 0 && d->res_x1 >= d->nx)) return false;
  const size_t strips = (d->ny + (size_t)(kWave - 1) * VecOf<T>::N - 1) / ((size_t)(kWave - 1) * VecOf<T>::N);
  // residual launches: one partial per wavefront, at best one chunk per (strip, plane), must fit the reduction workspace --
  // otherwise BackendPDHG takes the two-pass kernels (e.g. 1024^3, or ny = 4096 with 256 planes)
  if (strips * d->L > (size_t)kReduceBlocks / 2) return false;
  return strips * d->L * ((d->nx + 5) / 6) < (size_t)1 << 31;
}

template <class T>
static int run_iter3d(const prost_hip_fused_desc* d, T* x_new, T* y_new, const T* x, const T* y, const T* y_prev, double tau, double sigma, double theta,
                      int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* out4, void* ws, void* stream, void* record = nullptr,
                      const RuleTail* tail = nullptr) {
  if (!iter3d_ok<T>(d)) { set_error("fused 3-D iteration: unsupported description (see prost_hip_fused_iteration3d_supported)"); return 1; }
  if (!aligned16(x_new) || !aligned16(y_new) || !aligned16(x) || !aligned16(y) || !aligned16(y_prev)) { set_error("fused 3-D iteration: vectors must be 16-byte aligned"); return 1; }
  if (x_new == x || y_new == y || (out4 && y_new == y_prev)) { set_error("fused 3-D iteration: outputs must not alias inputs (planes l-1 / l+1 are read by other wavefronts)"); return 1; }
  if (out4 && (!ws || !y_prev)) { set_error("fused 3-D iteration: residuals need the reduction workspace and y_prev"); return 1; }
  constexpr int V = VecOf<T>::N;
  FusedArgs<T> a = make_fused_args<T>(d);
  const size_t strips = (d->ny + (size_t)(kWave - 1) * V - 1) / ((size_t)(kWave - 1) * V);
  size_t c = cols > 0 ? (size_t)cols : 18;
  if (cols <= 0) while (c > 6 && strips * d->L * ((d->nx + c - 1) / c) < 8192) c -= 3;     // enough wavefronts to fill the chip
  // residual launches: one partial of 4 doubles per wavefront must fit the reduction workspace
  const size_t max_waves = (size_t)kReduceBlocks / 2;
  if (out4) while (c < d->nx && strips * d->L * ((d->nx + c - 1) / c) > max_waves) c += 6;
  if (c > d->nx) c = d->nx;
  if (out4 && strips * d->L * ((d->nx + c - 1) / c) > max_waves) { set_error("fused 3-D iteration: grid exceeds the reduction workspace"); return 1; }
  a.cols_per_block = (unsigned)c;
  a.chunks = (unsigned)((d->nx + c - 1) / c);
  const unsigned grid = (unsigned)(strips * a.chunks * d->L);
  const UniformProx<T> ug = make_uniform_prox<T>(a.g_val, (T)tau * a.Tval);
  const UniformProx<T> uf = make_uniform_prox<T>(a.f_val, (T)sigma * a.Sval);
  hipStream_t s = as_stream(stream);
  double* partial = static_cast<double*>(ws);
  const PdhgRecord<T>* rec = static_cast<const PdhgRecord<T>*>(record);
  const bool gb = d->g_coeff_ptr[1] != nullptr;
  const bool gsq = d->g_fn == PROST_FN_SQUARE, fle = d->f_fn == PROST_FN_IND_LEQ0;
  const bool fast = gsq && fle && ug.a_one && ug.den_one && !ug.degenerate && a.g_val[3] == (T)0 && uf.a_one && uf.den_one && a.f_val[3] == (T)0;
#define GO2(G, F, B, FASTv, R) PH_LAUNCH((fused_iter3d_kernel<T, V, G, F, B, FASTv, R>), dim3(grid), dim3(kWave), 0, s, x_new, y_new, x, y, y_prev, a, (T)tau, (T)sigma, (T)theta, ug, uf, use_kty != 0, use_kx_prev != 0, use_kty_prev != 0, partial, rec)
#define GO(G, F, B, FASTv) do { if (out4) GO2(G, F, B, FASTv, true); else GO2(G, F, B, FASTv, false); } while (0)
  if (fast) { if (gb) GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, true, true); else GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, false, true); }
  else if (gsq && fle) { if (gb) GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, true, false); else GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, false, false); }
  else { if (gb) GO(-1, -1, true, false); else GO(-1, -1, false, false); }
#undef GO
#undef GO2
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "fused 3-D iteration kernel"); }
  if (out4 && tail && tail->apply) return launch_fold4_rule<T>(out4, partial, grid, record, tail->iteration, tail->mirror, s);
  if (out4) return launch_fold4(out4, partial, grid, s);
  return 0;
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_fused_iteration3d_supported(const prost_hip_fused_desc* d, int dtype) { return (dtype == 0 ? iter3d_ok<float>(d) : iter3d_ok<double>(d)) ? 1 : 0; }
int prost_hip_fused_iteration3d_f32(const prost_hip_fused_desc* d, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev,
                                    double tau, double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4,
                                    void* workspace, void* stream) {
  return run_iter3d<float>(d, x_new, y_new, x, y, y_prev, tau, sigma, theta, use_kty, use_kx_prev, use_kty_prev, cols, res_out4, workspace, stream);
}
int prost_hip_fused_iteration3d_f64(const prost_hip_fused_desc* d, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev,
                                    double tau, double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4,
                                    void* workspace, void* stream) {
  return run_iter3d<double>(d, x_new, y_new, x, y, y_prev, tau, sigma, theta, use_kty, use_kx_prev, use_kty_prev, cols, res_out4, workspace, stream);
}
// the same launch with tau, sigma, theta and the prox terms derived from them read from the device record (prost_hip_pdhg_rule_begin; the
// by-value structure flags -- which straight-line instance runs -- come from the description, they do not depend on the step sizes);
// apply_rule: the fold of a residual launch also evaluates the step-size rule and the stopping test (no communicator)
int prost_hip_fused_iteration3d_rec_f32(const prost_hip_fused_desc* d, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev, void* record,
                                        int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4, void* workspace, int apply_rule,
                                        unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!record) { set_error("fused_iteration3d_rec: no record"); return 1; }
  const RuleTail tail{apply_rule, iteration, mirror};
  return run_iter3d<float>(d, x_new, y_new, x, y, y_prev, 1.0, 1.0, 1.0, use_kty, use_kx_prev, use_kty_prev, cols, res_out4, workspace, stream, record, &tail);
}
int prost_hip_fused_iteration3d_rec_f64(const prost_hip_fused_desc* d, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev, void* record,
                                        int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4, void* workspace, int apply_rule,
                                        unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!record) { set_error("fused_iteration3d_rec: no record"); return 1; }
  const RuleTail tail{apply_rule, iteration, mirror};
  return run_iter3d<double>(d, x_new, y_new, x, y, y_prev, 1.0, 1.0, 1.0, use_kty, use_kx_prev, use_kty_prev, cols, res_out4, workspace, stream, record, &tail);
}
}  // extern "C"
