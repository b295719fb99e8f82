// kernels_fused_iter.hip -- ONE kernel per PDHG iteration for gradient2d problems.
//
// The two-pass scheme (kernels_fused.hip) moves 11 floats/pixel/iteration because x_new makes a
// round trip through HBM between the primal and the dual pass.  Here a wavefront marches over a
// chunk of columns and produces x_new[c+1] (primal step) one column AHEAD of y_new[c] (dual step),
// all in registers:
//     per column: load y1, y2, x, g-coefficients (f) of column c+1;  store x_new[c+1], y1_new[c], y2_new[c]
//     = 4 loads + 3 stores = 7 floats / pixel / iteration  (vs 11), bit-identical results.
// The dual step at column c needs x_new at (row+1, c) and (row, c+1):
//   * column c+1: the one-column software pipeline above;
//   * row+1: neighbour lane via a 64-lane shuffle.  Lane 63 of every wave is a HALO lane: it loads
//     and computes x_new for the 4 rows below the wave's 252 output rows (overlapping the next
//     wave's lane 0) but stores nothing -- no divergent halo code, 63/64 lane efficiency;
//   * the first column of a chunk is recomputed by the chunk on its left (1 halo column per chunk).
// Workgroup = one wavefront (64 lanes); grid = ceil(ny / 252) x column chunks, XCD-aware order.
// Loads for column c+2 are issued before the arithmetic of column c+1 (register prefetch), so each
// lane keeps >= 8 x 16 B requests in flight.
// RES = true (residual iterations) additionally streams y_prev (2 floats/pixel) and accumulates the
// four residual sums of backend_pdhg.cu:392-431 (one partial of 4 doubles per wavefront).
#include "pdhg_rule.hpp"
#include "reduce.hpp"


namespace prost_hip {

// GMASK: bit k set = g-coefficient k is a per-pixel vector (loaded with the column); the other
// coefficients are scalars that live in SGPRs.  ROF: only b = f is a vector -> GMASK = 0b10.
constexpr int popcount7(int m) { int c = 0; for (int k = 0; k < 7; k++) c += (m >> k) & 1; return c; }
constexpr int slot_of(int m, int k) { int c = 0; for (int i = 0; i < k; i++) c += (m >> i) & 1; return c; }

template <class T, int VEC, int LCH, int GMASK, bool RES>
struct ColIn {            // everything loaded for one column
  T y1[LCH][VEC], y2[LCH][VEC], x[LCH][VEC], gc[LCH][popcount7(GMASK) > 0 ? popcount7(GMASK) : 1][VEC];
  T up[LCH];              // y2 of the row above this wave's first row (only lane 0 loads it)
  T p1[RES ? LCH : 1][RES ? VEC : 1], p2[RES ? LCH : 1][RES ? VEC : 1], upp[RES ? LCH : 1];   // y_prev (RES only)
};

// RAG: the image height is not a multiple of VEC (fused_common.hpp, ldv_n / stv_n)
// FAST: straight-line instance for the ROF shape (host-checked: prox_g square with scalar a = 1, d = e = 0,
// prox_f* ind_leq0 with scalar a = 1, d = e = 0), same forms as kernels_fused_iter2.hip
// VART: the primal preconditioner depends on the position (FusedArgs::varT: the gradient handed over as the sparse matrix
// spmat_gradient2d, Tau_j = 1 / column sum).  Interior pixels (4 stencil entries in their column) run the code of the uniform
// instance with Tval = Tcls[2]; pixels of the first / last column and row -- 3 or 2 entries -- evaluate the reference's own
// expression with their Tau_j (elem_1d: ElemOperation1D as written; the straight-line instances: the divisor 1 + step_j of the pixel's
// class, EdgeTerms) and the residual terms with their sqrt(Tau_j).  The same for every channel.
// (launch bounds: the residual instance with position-dependent Tau would take 169-171 VGPRs -- two resident wavefronts per SIMD
// where the uniform instance, at 165-167, has three; it is held to three)
template <class T, int VEC, int LCH, int GFN, int FFN, int GMASK, bool RES, bool RAG, bool FAST, bool VART>
__global__ void __launch_bounds__(kWave, (RES && VART && FAST && LCH == 1) ? 3 : 1) fused_iter2d_kernel(T* __restrict__ x_new, T* __restrict__ y_new,
                                                             const T* __restrict__ x, const T* __restrict__ y,
                                                             const T* __restrict__ y_prev, FusedArgs<T> a, T tau, T sigma, T theta,
                                                             UniformProx<T> ug, UniformProx<T> uf, EdgeTerms<T> ec0, EdgeTerms<T> ec1,
                                                             bool use_kty, bool use_kx_prev, bool use_kty_prev,
                                                             double* __restrict__ partial, const PdhgRecord<T>* __restrict__ rec) {
  if (rec) {                       // device-resident step sizes (fused_common.hpp: PdhgRecord): wave-uniform scalar loads
    if (rec->stop) return;
    tau = rec->p.tau; sigma = rec->p.sigma; theta = rec->p.theta; ug = rec->p.ug; uf = rec->p.uf;
    if (VART) { ec0 = rec->p.ec[0]; ec1 = rec->p.ec[1]; }
  }
  const size_t nx = a.nx, ny = a.ny;
  const int lane = threadIdx.x;
  constexpr int kRowsPerWave = (kWave - 1) * VEC;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order; used for L2
  // locality only, never for correctness).  Each XCD gets a CONTIGUOUS range of tiles, and tiles
  // are ordered chunk-after-chunk inside a row strip, so the halo column a chunk shares with its
  // right neighbour is fetched by two workgroups of the same XCD close in time -> one L2 line.
  const unsigned total = gridDim.x, chunks = a.chunks;
  const unsigned xcd = blockIdx.x % 8u, q = blockIdx.x / 8u;
  const unsigned tile = xcd * (total / 8u) + (xcd < total % 8u ? xcd : total % 8u) + q;
  const unsigned strip = tile / chunks, chunk = tile % chunks;
  const size_t row0 = (size_t)strip * kRowsPerWave + (size_t)lane * VEC;
  const bool active = row0 < ny;                       // loads + primal step
  const bool owner = active && lane < kWave - 1;       // stores + residual terms
  const int nvalid = !RAG ? VEC : (active ? (ny - row0 < (size_t)VEC ? (int)(ny - row0) : VEC) : 0);   // rows of this lane inside the image
  const size_t xa = (size_t)chunk * a.cols_per_block;
  const size_t xb = xa + a.cols_per_block < nx ? xa + a.cols_per_block : nx;
  const size_t P = nx * ny, N = P * LCH;
  const T tauT = tau * a.Tval, sigS = sigma * a.Sval;
  const T sqT = t_sqrt(a.Tval), sqS = t_sqrt(a.Sval);
  const bool tiny_is_zero = a.f_val[1] >= (T)kTinyIsZeroRadius;     // device_math.hpp: norm2_leq0_fast
  // straight-line instance whose prox_f* is the Moreau wrap of ElemOperationNorm2<FFN> (FFN = abs: a problem written in the primal
  // form with the TV regulariser on the constrained variable, example_rof_primal.m:27): device_math.hpp, norm2_moreau_post
  constexpr bool kFM = FAST && FFN != PROST_FN_IND_LEQ0;
  const SharedDivisor<T> div_sS(kFM ? sigS : (T)1);                  // arg / (sigma Sigma), correctly rounded
  // Function1DSquare with scalar a, c, e: its fp64 divisor 1. + step is wave-uniform -> exact
  // reciprocal-based quotient instead of a ~35-instruction fp64 division per pixel (device_math.hpp)
  // scalar a, c, e of prox_g (always the case for prox_f* here): every element-independent part of
  // the scaled prox (fp64 denominators, step, reciprocal of Function1DSquare's divisor) was
  // evaluated on the host into ug / uf; see UniformProx in device_math.hpp
  constexpr bool kUniformG = (GMASK & 0x15) == 0;
  double r_pd = 0, r_pv = 0, r_dd = 0, r_dv = 0;       // primal diff^2, primal var^2, dual diff^2, dual var^2

  typedef ColIn<T, VEC, LCH, GMASK, RES> Col;
  // the residual divisors tau sqrt(T), sigma sqrt(S) are wave-uniform: exact quotients through one double reciprocal each
  const SharedDivisor<T> div_tauT(tau * sqT), div_sigS(sigma * sqS);
  auto load_col = [&](size_t c, Col& in) {
#pragma unroll
    for (int l = 0; l < LCH; l++) {
      const size_t o = l * P + c * ny + row0;
      ldv_n<T, VEC, RAG>(y + o, in.y1[l], nvalid); ldv_n<T, VEC, RAG>(y + N + o, in.y2[l], nvalid); ldv_n<T, VEC, RAG>(x + o, in.x[l], nvalid);
      // issued together with the column so that the primal step never waits on a second round trip
      in.up[l] = (lane == 0 && row0 > 0) ? y[N + o - 1] : (T)0;
      if (RES) {
        ldv_n<T, RES ? VEC : 1, RAG>(y_prev + o, in.p1[RES ? l : 0], nvalid);
        ldv_n<T, RES ? VEC : 1, RAG>(y_prev + N + o, in.p2[RES ? l : 0], nvalid);
        in.upp[RES ? l : 0] = (lane == 0 && row0 > 0) ? y_prev[N + o - 1] : (T)0;
      }
#pragma unroll
      for (int k = 0; k < 7; k++) {
        if ((GMASK >> k) & 1) {
          if (a.g_ptr[k]) ldv_n<T, VEC, RAG>(a.g_ptr[k] + o, in.gc[l][slot_of(GMASK, k)], nvalid);
          else {
#pragma unroll
            for (int j = 0; j < VEC; j++) in.gc[l][slot_of(GMASK, k)][j] = a.g_val[k];
          }
        }
      }
    }
  };
  // x_new of column c from its inputs and column c-1   (backend_pdhg.cu:317-338, block_gradient2d.cu:122-138)
  auto primal_col = [&](size_t c, const Col& in, const Col& prev, bool have_prev, bool owned, T (&xn)[LCH][VEC]) {
#pragma unroll
    for (int l = 0; l < LCH; l++) {
      T up = lane_up(in.y2[l][VEC - 1]);
      if (lane == 0) up = in.up[l];
      T upp = 0;
      if (RES) { upp = lane_up(in.p2[RES ? l : 0][RES ? VEC - 1 : 0]); if (lane == 0) upp = in.upp[RES ? l : 0]; }
      T ktyv[VEC], parg[VEC];
      T argv[VART ? VEC : 1], tTv[VART ? VEC : 1], sTv[VART ? VEC : 1];
      bool edgev[VART ? VEC : 1], cornerv[VART ? VEC : 1];
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const size_t row = row0 + j;
        T divy = (row < ny - 1) ? in.y2[l][j] : (T)0;
        if (row > 0) divy -= (j > 0 ? in.y2[l][j > 0 ? j - 1 : 0] : up);
        T divx = (c < nx - 1) ? in.y1[l][j] : (T)0;
        if (have_prev) divx -= prev.y1[l][j];
        T kty = use_kty ? (T)0 - (divx + divy) : (T)0;
        if (VART && use_kty) {
          // K^T y as the MATRIX's transpose sums it -- a CSR row of the stored transpose (csr_spmv: sum = 0; sum += val * x in column
          // order, i.e. d/dx rows before d/dy rows, the left / upper neighbour's +1 before the pixel's own -1): the association the
          // oracle's block.sparse uses, so that this path stays bit for bit with it (the stencil form above rounds differently)
          T s = 0;
          if (have_prev) s += prev.y1[l][j];
          if (c < nx - 1) s -= in.y1[l][j];
          if (row > 0) s += (j > 0 ? in.y2[l][j > 0 ? j - 1 : 0] : up);
          if (row < ny - 1) s -= in.y2[l][j];
          kty = s;
        }
        ktyv[j] = kty;
        T tT = tauT;
        bool edge = false;
        if (VART) {                // stencil entries in this pixel's column of K: 4 inside, 3 on an edge, 2 in a corner
          const int cnt = 4 - (c == 0 ? 1 : 0) - (c == nx - 1 ? 1 : 0) - (row == 0 ? 1 : 0) - (row == ny - 1 ? 1 : 0);
          edge = cnt != 4;
          const T Tj = cnt == 4 ? a.Tval : (cnt == 3 ? a.Tcls[1] : a.Tcls[0]);
          tT = tau * Tj;
          edgev[VART ? j : 0] = edge; cornerv[VART ? j : 0] = cnt == 2; tTv[VART ? j : 0] = tT; sTv[VART ? j : 0] = edge ? t_sqrt(Tj) : sqT;
        }
        const T arg = in.x[l][j] - tT * kty;
        if (VART) argv[VART ? j : 0] = arg;
        if (FAST) {
          parg[j] = arg - (((GMASK >> 1) & 1) ? in.gc[l][slot_of(GMASK, 1)][j] : a.g_val[1]);
        } else {
          T cf[7];
#pragma unroll
          for (int k = 0; k < 7; k++) cf[k] = ((GMASK >> k) & 1) ? in.gc[l][slot_of(GMASK, k)][j] : a.g_val[k];
          if (kUniformG && !edge) xn[l][j] = elem_1d_u<T, GFN>(a.g_fn, arg, cf, ug);
          else xn[l][j] = elem_1d<T, GFN>(a.g_fn, arg, tT, cf);
        }
      }
      if (FAST) {      // a (v - d tau) = v, fp64 denominator 1: F_prox(v - b; step) + b (square: exact division, one
        T r[VEC];      // fallback branch per vector; abs: soft threshold by step = c tau)
        if (GFN == PROST_FN_SQUARE) div_to_float_exact_vec<VEC>(parg, ug.sq, r);
        else {
#pragma unroll
          for (int j = 0; j < VEC; j++) r[j] = f1d_apply<T, GFN>(a.g_fn, parg[j], ug.step, a.g_val[5], a.g_val[6]);
        }
#pragma unroll
        for (int j = 0; j < VEC; j++) xn[l][j] = r[j] + (((GMASK >> 1) & 1) ? in.gc[l][slot_of(GMASK, 1)][j] : a.g_val[1]);
        if (VART) {
          // pixels with their own Tau_j: the same F_prox(v - b; step_j) + b with the step / the divisor 1 + step_j of the pixel's class
          // (EdgeTerms, formed once per launch) -- bit for bit what elem_1d evaluates for a = 1, d = e = 0
#pragma unroll
          for (int j = 0; j < VEC; j++) {
            if (edgev[VART ? j : 0]) {
              const bool cn = cornerv[VART ? j : 0];
              const T bj = ((GMASK >> 1) & 1) ? in.gc[l][slot_of(GMASK, 1)][j] : a.g_val[1];
              const T pj = argv[VART ? j : 0] - bj;
              T rj;
              if (GFN == PROST_FN_SQUARE) {
                UniformDiv dv;
                dv.D = cn ? ec0.sq.D : ec1.sq.D; dv.rD = cn ? ec0.sq.rD : ec1.sq.rD;
                rj = div_to_float_exact(pj, dv);
              } else rj = f1d_apply<T, GFN>(a.g_fn, pj, cn ? ec0.step : ec1.step, a.g_val[5], a.g_val[6]);
              xn[l][j] = rj + bj;
            }
          }
        }
      }
      if (RES) {                                            // dual_residual_transform (backend_pdhg.cu:73-94)
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          const size_t row = row0 + j;
          const int jj = RES ? j : 0, ll = RES ? l : 0;
          T dpy = (row < ny - 1) ? in.p2[ll][jj] : (T)0;
          if (row > 0) dpy -= (j > 0 ? in.p2[ll][RES && j > 0 ? j - 1 : 0] : upp);
          T dpx = (c < nx - 1) ? in.p1[ll][jj] : (T)0;
          if (have_prev) dpx -= prev.p1[ll][jj];
          T ktyp = use_kty_prev ? (T)0 - (dpx + dpy) : (T)0;
          if (VART && use_kty_prev) {          // the same sum in the matrix's order (see the primal step)
            T s = 0;
            if (have_prev) s += prev.p1[ll][jj];
            if (c < nx - 1) s -= in.p1[ll][jj];
            if (row > 0) s += (j > 0 ? in.p2[ll][RES && j > 0 ? j - 1 : 0] : upp);
            if (row < ny - 1) s -= in.p2[ll][jj];
            ktyp = s;
          }
          const T sT = VART ? sTv[VART ? j : 0] : sqT;
          const T w_hat = (VART && edgev[VART ? j : 0]) ? (in.x[l][j] - xn[l][j]) / (tau * sT) - sT * ktyp : div_tauT.div(in.x[l][j] - xn[l][j]) - sqT * ktyp;
          const T diff = w_hat + sT * ktyv[j];
          if (owner && owned && j < nvalid && c >= a.rx0 && c < a.rx1) { r_dd += (double)(diff * diff); r_dv += (double)(w_hat * w_hat); }
        }
      }
    }
  };

  Col cur = {}, nxt = {}, halo = {};
  T xn_c[LCH][VEC], xn_n[LCH][VEC];
#pragma unroll
  for (int l = 0; l < LCH; l++)
#pragma unroll
    for (int j = 0; j < VEC; j++) { xn_c[l][j] = 0; xn_n[l][j] = 0; }
  if (active) {
    load_col(xa, cur);
    if (xa > 0) {
#pragma unroll
      for (int l = 0; l < LCH; l++) {
        ldv_n<T, VEC, RAG>(y + l * P + (xa - 1) * ny + row0, halo.y1[l], nvalid);
        if (RES) ldv_n<T, RES ? VEC : 1, RAG>(y_prev + l * P + (xa - 1) * ny + row0, halo.p1[RES ? l : 0], nvalid);
      }
    }
    if (xa + 1 < nx) load_col(xa + 1, nxt);
  }
  primal_col(xa, cur, halo, xa > 0, true, xn_c);            // shuffles inside: every lane takes part
  if (owner) {
#pragma unroll
    for (int l = 0; l < LCH; l++) stv_n<T, VEC, true, RAG>(x_new + l * P + xa * ny + row0, xn_c[l], nvalid);
  }

  for (size_t c = xa; c < xb; c++) {
    const bool has_next = c + 1 < nx;
    Col pre;                                                  // prefetch column c+2 while column c+1 / c are processed
    const bool has_pre = c + 2 < nx && c + 1 < xb;
    if (active && has_pre) load_col(c + 2, pre);
    if (has_next) {
      primal_col(c + 1, nxt, cur, true, c + 1 < xb, xn_n);
      if (owner && c + 1 < xb) {
#pragma unroll
        for (int l = 0; l < LCH; l++) stv_n<T, VEC, true, RAG>(x_new + l * P + (c + 1) * ny + row0, xn_n[l], nvalid);
      }
    }
    // ---- dual step of column c (backend_pdhg.cu:341-370, block_gradient2d.cu:61-77) ----
    T bel_n[LCH], bel_o[LCH];
#pragma unroll
    for (int l = 0; l < LCH; l++) {
      bel_n[l] = lane_down(xn_c[l][0]);
      bel_o[l] = lane_down(cur.x[l][0]);
    }
    if (owner) {
      T out[2 * LCH][VEC];
      T av[FAST ? 2 * LCH : 1][FAST ? VEC : 1], nv[FAST ? VEC : 1];
      T vv[kFM ? 2 * LCH : 1][kFM ? VEC : 1];           // straight-line Moreau instance: the pre-scaled arguments
      T kxv[RES ? 2 * LCH : 1][RES ? VEC : 1], kpv[RES ? 2 * LCH : 1][RES ? VEC : 1];
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const size_t row = row0 + j;
        T arg[2 * LCH], kx[2 * LCH], kp[2 * LCH];
        T norm = 0;
#pragma unroll
        for (int l = 0; l < LCH; l++) {
          const T below_n = (j < VEC - 1) ? xn_c[l][j < VEC - 1 ? j + 1 : 0] : bel_n[l];
          const T below_o = (j < VEC - 1) ? cur.x[l][j < VEC - 1 ? j + 1 : 0] : bel_o[l];
          kx[l] = has_next ? xn_n[l][j] - xn_c[l][j] : (T)0;
          kx[LCH + l] = (row < ny - 1) ? below_n - xn_c[l][j] : (T)0;
          kp[l] = (use_kx_prev && has_next) ? nxt.x[l][j] - cur.x[l][j] : (T)0;
          kp[LCH + l] = (use_kx_prev && row < ny - 1) ? below_o - cur.x[l][j] : (T)0;
          arg[l] = cur.y1[l][j] + sigS * ((1 + theta) * kx[l] - theta * kp[l]);
          arg[LCH + l] = cur.y2[l][j] + sigS * ((1 + theta) * kx[LCH + l] - theta * kp[LCH + l]);
        }
        // prox_f* given as the Moreau wrap of prox_f (FusedArgs::fmor; prox_moreau.cu:98-134 with the dual call's invert_tau = false): the
        // operation sees v = arg / (sigma Sigma) and the step 1 / (sigma Sigma) (uf holds its terms), the result is arg - sigma Sigma r
        const bool fm = !FAST && a.fmor != 0;
        T va[2 * LCH];
#pragma unroll
        for (int i = 0; i < 2 * LCH; i++) va[i] = kFM ? div_sS.div(arg[i]) : (fm ? arg[i] / sigS : arg[i]);
#pragma unroll
        for (int i = 0; i < 2 * LCH; i++) norm += va[i] * va[i];
        if (RES) {
#pragma unroll
          for (int i = 0; i < 2 * LCH; i++) { kxv[RES ? i : 0][RES ? j : 0] = kx[i]; kpv[RES ? i : 0][RES ? j : 0] = kp[i]; }
        }
        if (FAST) {
          nv[FAST ? j : 0] = norm;
#pragma unroll
          for (int i = 0; i < 2 * LCH; i++) av[FAST ? i : 0][FAST ? j : 0] = arg[i];
          if (kFM) {
#pragma unroll
            for (int i = 0; i < 2 * LCH; i++) vv[kFM ? i : 0][kFM ? j : 0] = va[i];
          }
        } else if (norm > 0) {
          norm = t_sqrt(norm);
          const T pr = scaled_prox_u<T, FFN>(a.f_fn, norm, a.f_val, uf);
#pragma unroll
          for (int i = 0; i < 2 * LCH; i++) { const T r = pr * va[i] / norm; out[i][j] = fm ? arg[i] - sigS * r : r; }
        } else {
#pragma unroll
          for (int i = 0; i < 2 * LCH; i++) out[i][j] = fm ? arg[i] - sigS * (T)0 : (T)0;
        }
      }
      if constexpr (FAST) {
        // out = pr v / ||v||, pr = min(||v|| - b, 0) + b, 0 for ||v|| = 0: device_math.hpp
        if constexpr (kFM) norm2_moreau_post<T, FFN, kFM ? 2 * LCH : 1, kFM ? VEC : 1>(nv, vv, av, sigS, a.f_val, uf, out);
        else norm2_leq0_fast<T, 2 * LCH, VEC>(nv, av, a.f_val[1], tiny_is_zero, out);
      }
      if (RES && c >= a.rx0 && c < a.rx1) {                // primal_residual_transform (backend_pdhg.cu:97-120)
#pragma unroll
        for (int j = 0; j < VEC; j++) {
#pragma unroll
          for (int i = 0; i < 2 * LCH; i++) {
            const T yo = i < LCH ? cur.y1[i < LCH ? i : 0][j] : cur.y2[i < LCH ? 0 : i - LCH][j];
            const T kxi = kxv[RES ? i : 0][RES ? j : 0], kpi = kpv[RES ? i : 0][RES ? j : 0];
            const T z_hat = div_sigS.div(yo - out[i][j]) + sqS * ((1 + theta) * kxi - theta * kpi);
            const T diff = z_hat - sqS * kxi;
            if (j < nvalid) { r_pd += (double)(diff * diff); r_pv += (double)(z_hat * z_hat); }
          }
        }
      }
#pragma unroll
      for (int l = 0; l < LCH; l++) {
        stv_n<T, VEC, true, RAG>(y_new + l * P + c * ny + row0, out[l], nvalid);
        stv_n<T, VEC, true, RAG>(y_new + N + l * P + c * ny + row0, out[LCH + l], nvalid);
      }
    }
    // shift the pipeline
    cur = nxt;
    if (has_pre) nxt = pre;
#pragma unroll
    for (int l = 0; l < LCH; l++)
#pragma unroll
      for (int j = 0; j < VEC; j++) xn_c[l][j] = xn_n[l][j];
  }
  if (RES) {
    r_pd = wave_sum(r_pd); r_pv = wave_sum(r_pv); r_dd = wave_sum(r_dd); r_dv = wave_sum(r_dv);
    if (lane == 0) {
      double* p = partial + 4 * (size_t)blockIdx.x;
      p[0] = r_pd; p[1] = r_pv; p[2] = r_dd; p[3] = r_dv;
    }
  }
}

// a thread's share of the partials (slots t, t + kBlock, ...), added in that order; the loads of four slots are issued before the
// first add (a load-add loop pays one L2 latency per slot: 15 of them in front of the one workgroup at 4096^2) -- same order, same bits
__device__ __forceinline__ void fold4_accumulate(const double* __restrict__ partial, unsigned nslots, double (&v)[4]) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2* p2 = reinterpret_cast<const d2*>(partial);               // a slot = 4 doubles = two 16-byte loads (workspace: 256-byte aligned)
  for (unsigned i = threadIdx.x; i < nslots; i += 4 * kBlock) {
    d2 a[4][2];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const unsigned s = i + (unsigned)u * kBlock;
      if (s < nslots) { a[u][0] = p2[2 * (size_t)s]; a[u][1] = p2[2 * (size_t)s + 1]; }
      else { a[u][0] = d2{0, 0}; a[u][1] = d2{0, 0}; }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (i + (unsigned)u * kBlock < nslots) { v[0] += a[u][0][0]; v[1] += a[u][0][1]; v[2] += a[u][1][0]; v[3] += a[u][1][1]; }
  }
}

// folds nslots x 4 doubles in a fixed order (one workgroup) -> out4
__global__ void __launch_bounds__(kBlock) fold4_kernel(double* __restrict__ out4, const double* __restrict__ partial, unsigned nslots) {
  double v[4] = {0, 0, 0, 0};
  fold4_accumulate(partial, nslots, v);
  __shared__ double s[4][kBlock / kWave];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = wave_sum(v[k]);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) for (int k = 0; k < 4; k++) s[k][wave] = v[k];
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0;
    for (int w = 0; w < kBlock / kWave; w++) t += s[threadIdx.x][w];
    out4[threadIdx.x] = t;
  }
}

// the same fold with the step-size rule and the stopping test as its epilogue (pdhg_rule.hpp): one launch less per residual iteration
// where no all-reduce has to run between the sums and the rule
template <class T>
__global__ void __launch_bounds__(kBlock) fold4_rule_kernel(double* __restrict__ out4, const double* __restrict__ partial, unsigned nslots,
                                                            PdhgRecord<T>* rec, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror) {
  if (rec->stop) return;                 // the launch that would have produced the partials returned at once
  double v[4] = {0, 0, 0, 0};
  fold4_accumulate(partial, nslots, v);
  __shared__ double s[4][kBlock / kWave];
  __shared__ double tot[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = wave_sum(v[k]);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) for (int k = 0; k < 4; k++) s[k][wave] = v[k];
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0;
    for (int w = 0; w < kBlock / kWave; w++) t += s[threadIdx.x][w];
    out4[threadIdx.x] = t;
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) rule_apply_device(rec, tot, iteration, mirror);
}
template <class T>
int launch_fold4_rule(double* out4, const double* partial, unsigned nslots, void* rec, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, hipStream_t s) {
  hipLaunchKernelGGL(fold4_rule_kernel<T>, dim3(1), dim3(kBlock), 0, s, out4, partial, nslots, static_cast<PdhgRecord<T>*>(rec), iteration, mirror);
  PH_LAUNCH_END("fold4 + rule kernel");
}
template int launch_fold4_rule<float>(double*, const double*, unsigned, void*, unsigned long long, prost_hip_pdhg_rule_state*, hipStream_t);
template int launch_fold4_rule<double>(double*, const double*, unsigned, void*, unsigned long long, prost_hip_pdhg_rule_state*, hipStream_t);

int launch_fold4(double* out4, const double* partial, unsigned nslots, hipStream_t s) {
  hipLaunchKernelGGL(fold4_kernel, dim3(1), dim3(kBlock), 0, s, out4, partial, nslots);
  PH_LAUNCH_END("fold4 kernel");
}

static bool iter_desc_ok(const prost_hip_fused_desc* d, int dtype) {
  if (!d || d->is3d) return false;
  if (d->nx < 2 || d->ny < 2 || d->L < 1 || d->L > 2) return false;
  if (d->var_T && (d->nx < 4 || d->ny < 4)) return false;
  if (d->g_fn < 0 || d->g_fn >= PROST_FN_COUNT || d->f_fn < 0 || d->f_fn >= PROST_FN_COUNT) return false;
  const int V = dtype == 0 ? 4 : 2;
  for (int k = 0; k < 7; k++) {
    if (d->f_coeff_ptr[k]) return false;                 // per-pixel norm2 coefficients: two-pass kernels
    if (!aligned16(d->g_coeff_ptr[k])) return false;
  }
  const size_t strips = (d->ny + 63 * V - 1) / (63 * V);
  // residual launches write one partial (4 doubles) per wavefront into the reduction workspace: with ONE chunk per strip
  // (the widest choice the launcher can make) the strips alone must fit it, otherwise the two-pass kernels run
  return strips <= (size_t)kReduceBlocks / 2 && d->nx <= 65535 * 4;
}

template <class T>
static int run_iter(const prost_hip_fused_desc* d, T* x_new, T* y_new, const T* x, const T* y, const T* y_prev, double tau,
                    double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols, double* out4, void* ws,
                    void* stream, void* record = nullptr, const RuleTail* tail = nullptr) {
  constexpr int V = VecOf<T>::N;
  if (!iter_desc_ok(d, sizeof(T) == 4 ? 0 : 1) || !aligned16(x_new) || !aligned16(y_new) || !aligned16(x) || !aligned16(y) ||
      !aligned16(y_prev)) {
    set_error("fused iteration: unsupported description"); return 1;
  }
  if (out4 && (!ws || !y_prev)) { set_error("fused iteration: residuals need workspace and y_prev"); return 1; }
  FusedArgs<T> a = make_fused_args<T>(d);
  const size_t strips = (d->ny + 63 * V - 1) / (63 * V);
  if (cols <= 0) {
    // >= ~16 waves per CU (4096 on the chip), halo column <= 1/6 of a chunk.  Chunk lengths are kept
    // OFF powers of two: with ny a power of two a 16- or 32-column chunk puts every wave's streams
    // 2^k bytes apart and they collide on the same HBM channels (measured 4096^2 fp32: 16 cols
    // 0.121 ms, 12 or 18 cols 0.102 ms).
    cols = 18;
    while (cols > 6 && strips * ((d->nx + cols - 1) / cols) < 4096) cols -= 6;
    if (strips * ((d->nx + cols - 1) / cols) < 2048) {
      // small images: the launch lasts as long as one wave needs for its chunk (c + 1 steps); shortest
      // chunk that still fits one round of the 4096 wave slots
      double best = 1e30;
      for (int c : {6, 4, 3, 2, 1}) {
        const size_t waves = strips * ((d->nx + c - 1) / c);
        const double cost = (c + 1.5) * (double)((waves + 4095) / 4096);
        if (cost < best) { best = cost; cols = c; }
      }
    }
  }
  // residual launches write one partial (4 doubles) per wavefront: 2 * kReduceBlocks pairs fit the workspace
  while (out4 && (size_t)cols < d->nx && strips * ((d->nx + cols - 1) / cols) > (size_t)kReduceBlocks / 2) cols += 6;
  if (out4 && strips * ((d->nx + cols - 1) / cols) > (size_t)kReduceBlocks / 2) { set_error("fused iteration: grid exceeds the reduction workspace"); return 1; }
  a.cols_per_block = cols;
  a.chunks = (unsigned)((d->nx + cols - 1) / cols);
  if (strips * a.chunks > 0x7fffffffull) { set_error("fused iteration: grid too large"); return 1; }
  dim3 grid((unsigned)(strips * a.chunks)), block(kWave);
  hipStream_t s = as_stream(stream);
  double* partial = static_cast<double*>(ws);
  int mask = 0;
  for (int k = 0; k < 7; k++) if (d->g_coeff_ptr[k]) mask |= 1 << k;
  const bool rag = d->ny % V != 0;
  const PdhgRecord<T>* rec = static_cast<const PdhgRecord<T>*>(record);
  // with a device record the step sizes are not known here and the dispatch below may only depend on the coefficients: e = 0 on
  // both sides makes the fp64 denominators exactly 1 for every step size (what the straight-line instances assume)
  if (rec) { tau = sigma = theta = 1.0; }
  // host-side evaluation of everything element-independent, in the kernels' own expression order
  const UniformProx<T> ug = make_uniform_prox<T>(a.g_val, (T)tau * a.Tval);
  const UniformProx<T> uf = make_uniform_prox<T>(a.f_val, dual_prox_step<T>((T)sigma, a.Sval, a.fmor));
  const EdgeTerms<T> ec0 = a.varT ? make_edge_terms<T>(a.g_val, (T)tau * a.Tcls[0]) : EdgeTerms<T>(), ec1 = a.varT ? make_edge_terms<T>(a.g_val, (T)tau * a.Tcls[1]) : EdgeTerms<T>();
  // straight-line instance for the ROF shape (square / ind_leq0 with scalar a = 1, d = e = 0 on both sides); run-time
  // dispatched otherwise
  bool fast = (d->g_fn == PROST_FN_SQUARE || (d->g_fn == PROST_FN_ABS && d->L == 1 && mask == 0x2)) && d->f_fn == PROST_FN_IND_LEQ0 &&
              (mask == 0x2 || (mask == 0 && d->L == 1)) &&
              ug.a_one && ug.den_one && !ug.degenerate && a.g_val[3] == (T)0 && uf.a_one && uf.den_one && a.f_val[3] == (T)0;
  if (rec && (a.g_val[4] != (T)0 || a.f_val[4] != (T)0 || d->g_coeff_ptr[4])) fast = false;
  // Moreau-wrapped prox_f*: a straight-line instance for ElemOperationNorm2<abs> behind the square data term with per-pixel b (any
  // scalar coefficients: the operation's terms are in uf); the run-time dispatched instances otherwise
  const bool fast_g = d->g_fn == PROST_FN_SQUARE && mask == 0x2 && ug.a_one && ug.den_one && !ug.degenerate && a.g_val[3] == (T)0 &&
                      !(rec && (a.g_val[4] != (T)0 || d->g_coeff_ptr[4]));
  const bool fast_moreau = a.fmor && d->f_fn == PROST_FN_ABS && fast_g;
  if (a.fmor) fast = false;
#define GO3(LCHv, G, F, M, R, RAGv, FASTv, VARTv) PH_LAUNCH((fused_iter2d_kernel<T, V, LCHv, G, F, M, R, RAGv, FASTv, VARTv>), grid, block, 0, s, x_new, y_new, x, y, y_prev, a, (T)tau, (T)sigma, (T)theta, ug, uf, ec0, ec1, use_kty != 0, use_kx_prev != 0, use_kty_prev != 0, partial, rec)
#define GO2(LCHv, G, F, M, R, RAGv, FASTv) do { if (a.varT) GO3(LCHv, G, F, M, R, RAGv, FASTv, true); else GO3(LCHv, G, F, M, R, RAGv, FASTv, false); } while (0)
#define GO(LCHv, G, F, M, R, FASTv) do { if (rag) GO2(LCHv, G, F, M, R, true, FASTv); else GO2(LCHv, G, F, M, R, false, FASTv); } while (0)
#define GO_RES(LCHv, G, F, M, FASTv) do { if (out4) GO(LCHv, G, F, M, true, FASTv); else GO(LCHv, G, F, M, false, FASTv); } while (0)
  // measured (4096^2 fp32): non-temporal stores +4 %, non-temporal loads -15 %, no register prefetch -8 %
  if (d->L == 1) {
    if (fast_moreau) GO_RES(1, PROST_FN_SQUARE, PROST_FN_ABS, 0x2, true);                                // ROF in its primal form
    else if (fast && d->g_fn == PROST_FN_ABS) GO_RES(1, PROST_FN_ABS, PROST_FN_IND_LEQ0, 0x2, true);          // TV-L1 data term
    else if (fast && mask == 0x2) GO_RES(1, PROST_FN_SQUARE, PROST_FN_IND_LEQ0, 0x2, true);
    else if (fast) GO_RES(1, PROST_FN_SQUARE, PROST_FN_IND_LEQ0, 0, true);
    else if (mask == 0) GO_RES(1, -1, -1, 0, false);
    else GO_RES(1, -1, -1, 0x7F, false);
  } else {
    if (fast_moreau) GO_RES(2, PROST_FN_SQUARE, PROST_FN_ABS, 0x2, true);
    else if (fast) GO_RES(2, PROST_FN_SQUARE, PROST_FN_IND_LEQ0, 0x2, true); else if (mask == 0) GO_RES(2, -1, -1, 0, false); else GO_RES(2, -1, -1, 0x7F, false);
  }
#undef GO_RES
#undef GO
#undef GO2
#undef GO3
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "fused iteration kernel"); }
  if (out4 && tail && tail->apply) return launch_fold4_rule<T>(out4, partial, grid.x, record, tail->iteration, tail->mirror, s);
  if (out4) {
    hipLaunchKernelGGL(fold4_kernel, dim3(1), dim3(kBlock), 0, s, out4, partial, grid.x);
    PH_LAUNCH_END("fold4 kernel");
  }
  return 0;
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_fused_iteration_supported(const prost_hip_fused_desc* desc, int dtype) { return iter_desc_ok(desc, dtype) ? 1 : 0; }
int prost_hip_fused_iteration_f32(const prost_hip_fused_desc* d, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev,
                                  double tau, double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols_per_block,
                                  double* res_out4, void* workspace, void* s) {
  return run_iter<float>(d, x_new, y_new, x, y, y_prev, tau, sigma, theta, use_kty, use_kx_prev, use_kty_prev, cols_per_block, res_out4, workspace, s);
}
int prost_hip_fused_iteration_f64(const prost_hip_fused_desc* d, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev,
                                  double tau, double sigma, double theta, int use_kty, int use_kx_prev, int use_kty_prev, int cols_per_block,
                                  double* res_out4, void* workspace, void* s) {
  return run_iter<double>(d, x_new, y_new, x, y, y_prev, tau, sigma, theta, use_kty, use_kx_prev, use_kty_prev, cols_per_block, res_out4, workspace, s);
}
int prost_hip_fused_iteration_rec_f32(const prost_hip_fused_desc* d, float* x_new, float* y_new, const float* x, const float* y, const float* y_prev,
                                      void* record, int use_kty, int use_kx_prev, int use_kty_prev, int cols_per_block, double* res_out4, void* workspace,
                                      int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* s) {
  if (!record) { set_error("fused iteration: no step-size record"); return 1; }
  const RuleTail tail = {apply_rule, iteration, mirror};
  return run_iter<float>(d, x_new, y_new, x, y, y_prev, 1, 1, 1, use_kty, use_kx_prev, use_kty_prev, cols_per_block, res_out4, workspace, s, record, &tail);
}
int prost_hip_fused_iteration_rec_f64(const prost_hip_fused_desc* d, double* x_new, double* y_new, const double* x, const double* y, const double* y_prev,
                                      void* record, int use_kty, int use_kx_prev, int use_kty_prev, int cols_per_block, double* res_out4, void* workspace,
                                      int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* s) {
  if (!record) { set_error("fused iteration: no step-size record"); return 1; }
  const RuleTail tail = {apply_rule, iteration, mirror};
  return run_iter<double>(d, x_new, y_new, x, y, y_prev, 1, 1, 1, use_kty, use_kx_prev, use_kty_prev, cols_per_block, res_out4, workspace, s, record, &tail);
}
}  // extern "C"
