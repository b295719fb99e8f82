// kernels_fused_iter_mc.hip -- ONE kernel per PDHG iteration for gradient2d problems with L = 3 or 4 channels
// (vectorial TV of RGB images: example_rof_primaldual.m builds sum_norm2(2 * nc, ...) over all channel gradients).
//
// kernels_fused_iter.hip keeps all channels of a pixel in one lane (LCH <= 2); with three channels that is 289
// VGPRs and slower than the two passes.  Here the channels go ACROSS THE WAVEFRONTS OF A WORKGROUP: wavefront w
// marches over the same (row strip, column chunk) as its siblings but on channel w, with the single-channel
// register pipeline (x_new of column c+1 one step ahead of y_new of column c, row neighbours by wave shuffle,
// lane 63 a halo lane).  The only coupling between channels is the norm over the 2 L gradient components of a
// pixel: every wavefront writes the squares of its two dual arguments to LDS, one workgroup barrier per column,
// and every wavefront adds the 2 L squares in the reference's component order (all d/dx, then all d/dy; channel
// order inside) -- the same float additions as `norm += arg[i] * arg[i]`, so the result is bit-identical to the
// two-pass kernels and the oracle.  LDS: 2 buffers x 2 L components x 256 pixels x 4 B (6 KiB for RGB fp32),
// double-buffered so that one barrier per column suffices.
//     per channel and column: load y1, y2, x, f of column c+1; store x_new[c+1], y1_new[c], y2_new[c]
//     = 7 floats / pixel / channel / iteration (two passes: 11).
// RES = true (residual iterations) additionally streams y_prev of the channel and accumulates the four residual sums of
// backend_pdhg.cu:392-431 (one partial of 4 doubles per wavefront, folded by fold4).
#include "fused_common.hpp"
#include "reduce.hpp"

namespace prost_hip {

template <class T, int VEC, bool GB, bool RES>
struct ColMc {
  T y1[VEC], y2[VEC], x[VEC], b[GB ? VEC : 1];
  T up;              // y2 of the row above the wave's first row (lane 0)
  T p1[RES ? VEC : 1], p2[RES ? VEC : 1], pup;      // y_prev of this channel (RES only)
};

// GB: coefficient b of prox_g is a per-pixel vector; every other coefficient of prox_g and all of prox_f* are scalars.
// FAST: straight-line ROF instance (square / ind_leq0 with scalar a = 1, d = e = 0), forms of device_math.hpp.
// VART: position-dependent primal preconditioner (FusedArgs::varT; see fused_iter2d_kernel in kernels_fused_iter.hip) -- the same for
// every channel: spmat_gradient2d(nx, ny, L) repeats the one-channel matrix L times.
template <class T, int VEC, int GFN, int FFN, bool GB, bool FAST, int LW, bool RES, bool VART>
__global__ void __launch_bounds__(kWave * LW) fused_iter2d_mc_kernel(T* __restrict__ x_new, T* __restrict__ y_new, const T* __restrict__ x,
                                                                    const T* __restrict__ y, const T* __restrict__ y_prev, FusedArgs<T> a, T tau,
                                                                    T sigma, T theta, UniformProx<T> ug, UniformProx<T> uf, EdgeTerms<T> ec0, EdgeTerms<T> ec1,
                                                                    bool use_kty, bool use_kx_prev, bool use_kty_prev, double* __restrict__ partial,
                                                                    const PdhgRecord<T>* __restrict__ rec) {
  if (rec) {                       // device-resident step sizes (fused_common.hpp: PdhgRecord): wave-uniform scalar loads
    if (rec->stop) return;
    tau = rec->p.tau; sigma = rec->p.sigma; theta = rec->p.theta; ug = rec->p.ug; uf = rec->p.uf;
    if (VART) { ec0 = rec->p.ec[0]; ec1 = rec->p.ec[1]; }
  }
  constexpr int kRowsPerWave = (kWave - 1) * VEC;
  constexpr int kPix = kWave * VEC;
  __shared__ T s_sq[2][2 * LW][kPix];
  const size_t nx = a.nx, ny = a.ny;
  const int lane = threadIdx.x & (kWave - 1), ch = threadIdx.x / kWave;
  const unsigned total = gridDim.x, chunks = a.chunks;
  const unsigned xcd = blockIdx.x % 8u, q = blockIdx.x / 8u;            // XCD-aware tile order, see kernels_fused_iter.hip
  const unsigned tile = xcd * (total / 8u) + (xcd < total % 8u ? xcd : total % 8u) + q;
  const unsigned strip = tile / chunks, chunk = tile % chunks;
  const size_t row0 = (size_t)strip * kRowsPerWave + (size_t)lane * VEC;
  const bool active = row0 < ny;
  const bool owner = active && lane < kWave - 1;
  const size_t xa = (size_t)chunk * a.cols_per_block;
  const size_t xb = xa + a.cols_per_block < nx ? xa + a.cols_per_block : nx;
  const size_t P = nx * ny, N = P * LW, plane = (size_t)ch * P;
  const T tauT = tau * a.Tval, sigS = sigma * a.Sval;
  // straight-line instance whose prox_f* is the Moreau wrap of ElemOperationNorm2<FFN> (kernels_fused_iter.hip; device_math.hpp: norm2_moreau_post)
  constexpr bool kFM = FAST && FFN != PROST_FN_IND_LEQ0, kPlain = FAST && !kFM;
  const SharedDivisor<T> div_sS(kFM ? sigS : (T)1);       // arg / (sigma Sigma), correctly rounded
  const T* y1 = y + plane; const T* y2 = y + N + plane;
  const T* xp = x + plane;
  const T* bp = GB ? a.g_ptr[1] + plane : nullptr;
  const T* q1 = RES ? y_prev + plane : nullptr; const T* q2 = RES ? y_prev + N + plane : nullptr;
  const T sqT = t_sqrt(a.Tval), sqS = t_sqrt(a.Sval);
  const bool tiny_is_zero = a.f_val[1] >= (T)kTinyIsZeroRadius;     // device_math.hpp: norm2_leq0_fast
  const SharedDivisor<T> div_tauT(tau * sqT), div_sigS(sigma * sqS);        // wave-uniform residual divisors
  double r_pd = 0, r_pv = 0, r_dd = 0, r_dv = 0;       // primal diff^2, primal var^2, dual diff^2, dual var^2

  typedef ColMc<T, VEC, GB, RES> Col;
  auto load_col = [&](size_t c, Col& in) {
    const size_t o = c * ny + row0;
    ldv<T, VEC>(y1 + o, in.y1); ldv<T, VEC>(y2 + o, in.y2); ldv<T, VEC>(xp + o, in.x);
    if (GB) ldv<T, GB ? VEC : 1>(bp + o, in.b);
    in.up = (lane == 0 && row0 > 0) ? y2[o - 1] : (T)0;
    if (RES) {
      ldv<T, RES ? VEC : 1>(q1 + o, in.p1); ldv<T, RES ? VEC : 1>(q2 + o, in.p2);
      in.pup = (lane == 0 && row0 > 0) ? q2[o - 1] : (T)0;
    }
  };
  // x_new of this channel at column c (backend_pdhg.cu:317-338, block_gradient2d.cu:122-138 on a zero-filled result)
  auto primal_col = [&](size_t c, const Col& in, const T (&p1)[VEC], bool have_prev, T (&xn)[VEC], const T (&pp1)[RES ? VEC : 1], bool counted) {
    T up = lane_up(in.y2[VEC - 1]);
    if (lane == 0) up = in.up;
    T parg[VEC], ktyv[RES ? VEC : 1];
    T argv[VART ? VEC : 1], sTv[VART ? VEC : 1];
    bool edgev[VART ? VEC : 1], cornerv[VART ? VEC : 1];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const size_t row = row0 + j;
      T divy = (row < ny - 1) ? in.y2[j] : (T)0;
      if (row > 0) divy -= (j > 0 ? in.y2[j > 0 ? j - 1 : 0] : up);
      T divx = (c < nx - 1) ? in.y1[j] : (T)0;
      if (have_prev) divx -= p1[j];
      T kty = use_kty ? (T)0 - (divx + divy) : (T)0;
      if (VART && use_kty) {         // K^T y in the order of the matrix's transposed CSR row (kernels_fused_iter.hip)
        T s = 0;
        if (have_prev) s += p1[j];
        if (c < nx - 1) s -= in.y1[j];
        if (row > 0) s += (j > 0 ? in.y2[j > 0 ? j - 1 : 0] : up);
        if (row < ny - 1) s -= in.y2[j];
        kty = s;
      }
      if (RES) ktyv[RES ? j : 0] = kty;
      T tT = tauT;
      bool edge = false;
      if (VART) {                  // stencil entries in this pixel's column of K: 4 inside, 3 on an edge, 2 in a corner
        const int cnt = 4 - (c == 0 ? 1 : 0) - (c == nx - 1 ? 1 : 0) - (row == 0 ? 1 : 0) - (row == ny - 1 ? 1 : 0);
        edge = cnt != 4;
        const T Tj = cnt == 4 ? a.Tval : (cnt == 3 ? a.Tcls[1] : a.Tcls[0]);
        tT = tau * Tj;
        edgev[VART ? j : 0] = edge; cornerv[VART ? j : 0] = cnt == 2; sTv[VART ? j : 0] = edge ? t_sqrt(Tj) : sqT;
      }
      const T arg = in.x[j] - tT * kty;
      if (VART) argv[VART ? j : 0] = arg;
      if (FAST) {
        parg[j] = arg - (GB ? in.b[GB ? j : 0] : a.g_val[1]);
      } else {
        T cf[7];
#pragma unroll
        for (int k = 0; k < 7; k++) cf[k] = a.g_val[k];
        if (GB) cf[1] = in.b[GB ? j : 0];
        if (!edge) xn[j] = elem_1d_u<T, GFN>(a.g_fn, arg, cf, ug);
        else xn[j] = elem_1d<T, GFN>(a.g_fn, arg, tT, cf);
      }
    }
    if (FAST) {
      T r[VEC];
      div_to_float_exact_vec<VEC>(parg, ug.sq, r);
#pragma unroll
      for (int j = 0; j < VEC; j++) xn[j] = r[j] + (GB ? in.b[GB ? j : 0] : a.g_val[1]);
      if (VART) {                  // pixels with their own Tau_j: the divisor 1 + step_j of their class (EdgeTerms)
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          if (edgev[VART ? j : 0]) {
            const bool cn = cornerv[VART ? j : 0];
            UniformDiv dv;
            dv.D = cn ? ec0.sq.D : ec1.sq.D; dv.rD = cn ? ec0.sq.rD : ec1.sq.rD;
            const T bj = GB ? in.b[GB ? j : 0] : a.g_val[1];
            xn[j] = div_to_float_exact(argv[VART ? j : 0] - bj, dv) + bj;
          }
        }
      }
    }
    if (RES) {                                              // dual_residual_transform (backend_pdhg.cu:73-94)
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
        T ktyp = use_kty_prev ? (T)0 - (dpx + dpy) : (T)0;
        if (VART && use_kty_prev) {
          T s = 0;
          if (have_prev) s += pp1[jj];
          if (c < nx - 1) s -= in.p1[jj];
          if (row > 0) s += (j > 0 ? in.p2[RES && j > 0 ? j - 1 : 0] : upp);
          if (row < ny - 1) s -= in.p2[jj];
          ktyp = s;
        }
        const T sT = VART ? sTv[VART ? j : 0] : sqT;
        const T w_hat = (VART && edgev[VART ? j : 0]) ? (in.x[j] - xn[j]) / (tau * sT) - sT * ktyp : div_tauT.div(in.x[j] - xn[j]) - sqT * ktyp;
        const T diff = w_hat + sT * ktyv[jj];
        if (owner && counted && c >= a.rx0 && c < a.rx1) { r_dd += (double)(diff * diff); r_dv += (double)(w_hat * w_hat); }
      }
    }
  };

  Col cur = {}, nxt = {};
  T h1[VEC], xn_c[VEC], xn_n[VEC], hp1[RES ? VEC : 1];
#pragma unroll
  for (int j = 0; j < VEC; j++) { h1[j] = 0; xn_c[j] = 0; xn_n[j] = 0; }
#pragma unroll
  for (int j = 0; j < (RES ? VEC : 1); j++) hp1[j] = 0;
  if (active) {
    load_col(xa, cur);
    if (xa > 0) {
      ldv<T, VEC>(y1 + (xa - 1) * ny + row0, h1);
      if (RES) ldv<T, RES ? VEC : 1>(q1 + (xa - 1) * ny + row0, hp1);
    }
    if (xa + 1 < nx) load_col(xa + 1, nxt);
  }
  primal_col(xa, cur, h1, xa > 0, xn_c, hp1, true);
  if (owner) stv_nt<T, VEC>(x_new + plane + xa * ny + row0, xn_c);

  for (size_t c = xa; c < xb; c++) {
    const bool has_next = c + 1 < nx;
    Col pre;
    const bool has_pre = c + 2 < nx && c + 1 < xb;
    if (active && has_pre) load_col(c + 2, pre);
    if (has_next) {
      primal_col(c + 1, nxt, cur.y1, true, xn_n, cur.p1, c + 1 < xb);
      if (owner && c + 1 < xb) stv_nt<T, VEC>(x_new + plane + (c + 1) * ny + row0, xn_n);
    }
    // ---- dual step of column c (backend_pdhg.cu:341-370, block_gradient2d.cu:61-77) ----
    const T bel_n = lane_down(xn_c[0]);
    const T bel_o = lane_down(cur.x[0]);
    T av[2][VEC];
    T vv[kPlain ? 1 : 2][kPlain ? 1 : VEC];                // what the norm2 operation sees (the arguments, or their Moreau pre-scaled form)
    const bool fm = !FAST && a.fmor != 0;
    T kxv[RES ? 2 : 1][RES ? VEC : 1], kpv[RES ? 2 : 1][RES ? VEC : 1];
    const int buf = (int)(c & 1);
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const size_t row = row0 + j;
      const T below_n = (j < VEC - 1) ? xn_c[j < VEC - 1 ? j + 1 : 0] : bel_n;
      const T below_o = (j < VEC - 1) ? cur.x[j < VEC - 1 ? j + 1 : 0] : bel_o;
      const T kx0 = has_next ? xn_n[j] - xn_c[j] : (T)0;
      const T kx1 = (row < ny - 1) ? below_n - xn_c[j] : (T)0;
      const T kp0 = (use_kx_prev && has_next) ? nxt.x[j] - cur.x[j] : (T)0;
      const T kp1 = (use_kx_prev && row < ny - 1) ? below_o - cur.x[j] : (T)0;
      av[0][j] = cur.y1[j] + sigS * ((1 + theta) * kx0 - theta * kp0);       // backend_pdhg.cu:54-70
      av[1][j] = cur.y2[j] + sigS * ((1 + theta) * kx1 - theta * kp1);
      if (RES) { kxv[0][RES ? j : 0] = kx0; kxv[RES ? 1 : 0][RES ? j : 0] = kx1; kpv[0][RES ? j : 0] = kp0; kpv[RES ? 1 : 0][RES ? j : 0] = kp1; }
      // prox_f* given as the Moreau wrap of prox_f (FusedArgs::fmor; prox_moreau.cu:98-134 with the dual call's invert_tau = false): the
      // operation sees v = arg / (sigma Sigma) and the step 1 / (sigma Sigma) (uf holds its terms), the result is arg - sigma Sigma r
      if (kFM) { vv[0][kPlain ? 0 : j] = div_sS.div(av[0][j]); vv[kPlain ? 0 : 1][kPlain ? 0 : j] = div_sS.div(av[1][j]); }
      else if (!FAST) { vv[0][kPlain ? 0 : j] = fm ? av[0][j] / sigS : av[0][j]; vv[kPlain ? 0 : 1][kPlain ? 0 : j] = fm ? av[1][j] / sigS : av[1][j]; }
      const T w0 = kPlain ? av[0][j] : vv[0][kPlain ? 0 : j], w1 = kPlain ? av[1][j] : vv[kPlain ? 0 : 1][kPlain ? 0 : j];
      s_sq[buf][ch][j * kWave + lane] = w0 * w0;                         // [j][lane]: conflict-free banks
      s_sq[buf][LW + ch][j * kWave + lane] = w1 * w1;
    }
    __syncthreads();                                        // every wavefront of the workgroup runs the same column loop
    T nv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      T norm = 0;
#pragma unroll
      for (int i = 0; i < 2 * LW; i++) norm += s_sq[buf][i][j * kWave + lane];   // component order of ElemOperationNorm2: d/dx of all channels, then d/dy
      nv[j] = norm;
    }
    if (owner) {
      T out[2][VEC];
      if constexpr (kFM) {
        norm2_moreau_post<T, FFN, 2, VEC>(nv, vv, av, sigS, a.f_val, uf, out);
      } else if constexpr (FAST) {
        norm2_leq0_fast<T, 2, VEC>(nv, av, a.f_val[1], tiny_is_zero, out);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          if (nv[j] > 0) {
            const T nrm = t_sqrt(nv[j]);
            const T pr = scaled_prox_u<T, FFN>(a.f_fn, nrm, a.f_val, uf);
#pragma unroll
            for (int i = 0; i < 2; i++) { const T r = pr * vv[kPlain ? 0 : i][kPlain ? 0 : j] / nrm; out[i][j] = fm ? av[i][j] - sigS * r : r; }
          } else {
#pragma unroll
            for (int i = 0; i < 2; i++) out[i][j] = fm ? av[i][j] - sigS * (T)0 : (T)0;
          }
        }
      }
      if (RES && c >= a.rx0 && c < a.rx1) {                  // primal_residual_transform (backend_pdhg.cu:97-120); owned columns only
#pragma unroll
        for (int j = 0; j < VEC; j++) {
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const T yo = i == 0 ? cur.y1[j] : cur.y2[j];
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
    }
    cur = nxt;
    if (has_pre) nxt = pre;
#pragma unroll
    for (int j = 0; j < VEC; j++) xn_c[j] = xn_n[j];
  }
  if (RES) {
    r_pd = wave_sum(r_pd); r_pv = wave_sum(r_pv); r_dd = wave_sum(r_dd); r_dv = wave_sum(r_dv);
    if (lane == 0) {
      double* p = partial + 4 * ((size_t)blockIdx.x * LW + ch);
      p[0] = r_pd; p[1] = r_pv; p[2] = r_dd; p[3] = r_dv;
    }
  }
}

template <class T>
static bool iter_mc_ok(const prost_hip_fused_desc* d) {
  if (!d || d->is3d || (d->L != 3 && d->L != 4)) return false;
  if (d->nx == 0 || d->ny == 0) return false;
  if (d->var_T && (d->nx < 4 || d->ny < 4)) return false;
  if (d->g_fn < 0 || d->g_fn >= PROST_FN_COUNT || d->f_fn < 0 || d->f_fn >= PROST_FN_COUNT) return false;
  for (int k = 0; k < 7; k++) {
    if (d->f_coeff_ptr[k]) return false;
    if (k != 1 && d->g_coeff_ptr[k]) return false;
  }
  if (d->g_coeff_ptr[1] && !aligned16(d->g_coeff_ptr[1])) return false;
  // (heights that are not a whole number of 16-byte row groups run the one-row-per-lane instance: round 3)
  const size_t vec = d->ny % VecOf<T>::N == 0 ? VecOf<T>::N : 1;
  const size_t strips = (d->ny + (size_t)(kWave - 1) * vec - 1) / ((size_t)(kWave - 1) * vec);
  if (strips * d->L > (size_t)kReduceBlocks / 2) return false;     // residual launches: one partial per wavefront must fit the workspace
  return strips * d->nx < (size_t)1 << 31;
}

template <class T, int V>
static int run_iter_mc_v(const prost_hip_fused_desc* d, T* x_new, T* y_new, const T* x, const T* y, const T* y_prev, double tau, double sigma, double theta,
                       int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* out4, void* ws, void* stream, void* record = nullptr, const RuleTail* tail = nullptr) {
  if (!iter_mc_ok<T>(d)) { set_error("fused multi-channel iteration: unsupported description (see prost_hip_fused_iteration_mc_supported)"); return 1; }
  if (!aligned16(x_new) || !aligned16(y_new) || !aligned16(x) || !aligned16(y) || !aligned16(y_prev)) { set_error("fused multi-channel iteration: vectors must be 16-byte aligned"); return 1; }
  if (x_new == x || y_new == y || (out4 && y_new == y_prev)) { set_error("fused multi-channel iteration: outputs must not alias inputs"); return 1; }
  if (out4 && (!ws || !y_prev)) { set_error("fused multi-channel iteration: residuals need the reduction workspace and y_prev"); return 1; }
  FusedArgs<T> a = make_fused_args<T>(d);
  const size_t strips = (d->ny + (size_t)(kWave - 1) * V - 1) / ((size_t)(kWave - 1) * V);
  // measured 4096^2 x 3 fp32: 6 columns 0.302 ms, 12: 0.306, 18: 0.314, 36: 0.329 (shorter chunks = more workgroups in flight
  // to hide the per-column barrier; one halo column each)
  size_t c = cols > 0 ? (size_t)cols : 12;
  if (cols <= 0) while (c > 3 && strips * ((d->nx + c - 1) / c) * d->L < 4096) c -= 3;
  // residual launches: one partial of 4 doubles per wavefront must fit the reduction workspace
  const size_t max_waves = (size_t)kReduceBlocks / 2;
  if (out4) while (c < d->nx && strips * ((d->nx + c - 1) / c) * d->L > max_waves) c += 6;
  if (c > d->nx) c = d->nx;
  if (out4 && strips * ((d->nx + c - 1) / c) * d->L > max_waves) { set_error("fused multi-channel iteration: grid exceeds the reduction workspace"); return 1; }
  a.cols_per_block = (unsigned)c;
  a.chunks = (unsigned)((d->nx + c - 1) / c);
  const unsigned grid = (unsigned)(strips * a.chunks);
  const PdhgRecord<T>* rec = static_cast<const PdhgRecord<T>*>(record);
  if (rec) { tau = sigma = theta = 1.0; }      // the dispatch below may only depend on the coefficients (kernels_fused_iter.hip: run_iter)
  const UniformProx<T> ug = make_uniform_prox<T>(a.g_val, (T)tau * a.Tval);
  const UniformProx<T> uf = make_uniform_prox<T>(a.f_val, dual_prox_step<T>((T)sigma, a.Sval, a.fmor));
  hipStream_t s = as_stream(stream);
  double* partial = static_cast<double*>(ws);
  const bool gb = d->g_coeff_ptr[1] != nullptr;
  const bool gsq = d->g_fn == PROST_FN_SQUARE, fle = d->f_fn == PROST_FN_IND_LEQ0;
  bool fast = gsq && fle && ug.a_one && ug.den_one && !ug.degenerate && a.g_val[3] == (T)0 && uf.a_one && uf.den_one && a.f_val[3] == (T)0;
  if (rec && (a.g_val[4] != (T)0 || a.f_val[4] != (T)0)) fast = false;
  // Moreau-wrapped prox_f*: a straight-line instance for ElemOperationNorm2<abs> behind the square data term with per-pixel b
  // (kernels_fused_iter.hip: run_iter); the run-time dispatched instances otherwise
  const bool fast_moreau = a.fmor && d->f_fn == PROST_FN_ABS && gb && gsq && ug.a_one && ug.den_one && !ug.degenerate && a.g_val[3] == (T)0 && !(rec && a.g_val[4] != (T)0);
  if (a.fmor) fast = false;
  const EdgeTerms<T> ec0 = a.varT ? make_edge_terms<T>(a.g_val, (T)tau * a.Tcls[0]) : EdgeTerms<T>(), ec1 = a.varT ? make_edge_terms<T>(a.g_val, (T)tau * a.Tcls[1]) : EdgeTerms<T>();
#define GO4(G, F, B, FASTv, LWv, R, VARTv) PH_LAUNCH((fused_iter2d_mc_kernel<T, V, G, F, B, FASTv, LWv, R, VARTv>), dim3(grid), dim3(kWave * LWv), 0, s, x_new, y_new, x, y, y_prev, a, (T)tau, (T)sigma, (T)theta, ug, uf, ec0, ec1, use_kty != 0, use_kx_prev != 0, use_kty_prev != 0, partial, rec)
#define GO3(G, F, B, FASTv, LWv, R) do { if (a.varT) GO4(G, F, B, FASTv, LWv, R, true); else GO4(G, F, B, FASTv, LWv, R, false); } while (0)
#define GO2(G, F, B, FASTv, LWv) do { if (out4) GO3(G, F, B, FASTv, LWv, true); else GO3(G, F, B, FASTv, LWv, false); } while (0)
#define GO(G, F, B, FASTv) do { if (d->L == 3) GO2(G, F, B, FASTv, 3); else GO2(G, F, B, FASTv, 4); } while (0)
  if (fast_moreau) GO(PROST_FN_SQUARE, PROST_FN_ABS, true, true);
  else if (fast) { if (gb) GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, true, true); else GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, false, true); }
  else { if (gb) GO(-1, -1, true, false); else GO(-1, -1, false, false); }
#undef GO
#undef GO2
#undef GO3
#undef GO4
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "fused multi-channel iteration kernel"); }
  if (out4 && tail && tail->apply) return launch_fold4_rule<T>(out4, partial, grid * (unsigned)d->L, record, tail->iteration, tail->mirror, s);
  if (out4) return launch_fold4(out4, partial, grid * (unsigned)d->L, s);
  return 0;
}

// 16 bytes of rows per lane where the height is a whole number of such groups, one row per lane otherwise
template <class T>
static int run_iter_mc(const prost_hip_fused_desc* d, T* x_new, T* y_new, const T* x, const T* y, const T* y_prev, double tau, double sigma, double theta,
                       int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* out4, void* ws, void* stream, void* record = nullptr, const RuleTail* tail = nullptr) {
  if (d && d->ny % VecOf<T>::N != 0) return run_iter_mc_v<T, 1>(d, x_new, y_new, x, y, y_prev, tau, sigma, theta, use_kty, use_kx_prev, use_kty_prev, cols, out4, ws, stream, record, tail);
  return run_iter_mc_v<T, VecOf<T>::N>(d, x_new, y_new, x, y, y_prev, tau, sigma, theta, use_kty, use_kx_prev, use_kty_prev, cols, out4, ws, stream, record, tail);
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_fused_iteration_mc_supported(const prost_hip_fused_desc* d, int dtype) { return (dtype == 0 ? iter_mc_ok<float>(d) : iter_mc_ok<double>(d)) ? 1 : 0; }
int prost_hip_fused_iteration_mc_f32(const prost_hip_fused_desc* d, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev, double tau,
                                     double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4, void* workspace,
                                     void* stream) {
  return run_iter_mc<float>(d, x_new, y_new, x, y, y_prev, tau, sigma, theta, use_kty, use_kx_prev, use_kty_prev, cols, res_out4, workspace, stream);
}
int prost_hip_fused_iteration_mc_f64(const prost_hip_fused_desc* d, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev, double tau,
                                     double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4, void* workspace,
                                     void* stream) {
  return run_iter_mc<double>(d, x_new, y_new, x, y, y_prev, tau, sigma, theta, use_kty, use_kx_prev, use_kty_prev, cols, res_out4, workspace, stream);
}
int prost_hip_fused_iteration_mc_rec_f32(const prost_hip_fused_desc* d, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev,
                                         void* record, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4, void* workspace,
                                         int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!record) { set_error("fused multi-channel iteration: no step-size record"); return 1; }
  const RuleTail tail = {apply_rule, iteration, mirror};
  return run_iter_mc<float>(d, x_new, y_new, x, y, y_prev, 1, 1, 1, use_kty, use_kx_prev, use_kty_prev, cols, res_out4, workspace, stream, record, &tail);
}
int prost_hip_fused_iteration_mc_rec_f64(const prost_hip_fused_desc* d, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev,
                                         void* record, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* res_out4, void* workspace,
                                         int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!record) { set_error("fused multi-channel iteration: no step-size record"); return 1; }
  const RuleTail tail = {apply_rule, iteration, mirror};
  return run_iter_mc<double>(d, x_new, y_new, x, y, y_prev, 1, 1, 1, use_kty, use_kx_prev, use_kty_prev, cols, res_out4, workspace, stream, record, &tail);
}
}  // extern "C"
