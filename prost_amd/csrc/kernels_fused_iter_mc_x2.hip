// kernels_fused_iter_mc_x2.hip -- gradient2d with L = 2, 3 or 4 channels (vectorial TV: RGB images, flow fields), TWO PDHG
// iterations per kernel launch.
//
// The channels run on the wavefronts of a workgroup as in kernels_fused_iter_mc.hip; every wavefront runs the 4-stage column
// pipeline of the gray-value pair kernel (kernels_fused_iter2.hip) on ITS channel:
//     A(c+2): x1 = primal step of iteration k            B(c+1): y1 = dual step of iteration k
//     C(c)  : x2 = primal step of iteration k+1          D(c-1): y2 = dual step of iteration k+1
// The only coupling between channels is the norm over the 2 L gradient components of a pixel in the two dual steps: stages B
// and D first form their dual arguments and publish the squares in LDS, ONE workgroup barrier per column step, then every
// wavefront adds the 2 L squares of both stages in the reference's component order (all d/dx, then all d/dy; channel order
// inside: the float additions of `norm += arg[i] * arg[i]`) and finishes the two projections.  Double-buffered LDS
// (2 x 2 stages x 2 L x 256 floats = 32 KiB for L = 4), so that the barrier of the next step is the only other fence.
// Row neighbours come from adjacent lanes (DPP), lanes 0 and 63 are halo lanes as in the gray-value pair kernel.
// Per channel: 7 floats per pixel per TWO iterations (y1, y2, x, b read; x, y1, y2 written) against 2 x 7 for the
// single-iteration kernel.  x^(k+2), y^(k+2) are bit-identical to two single launches (tests/test_gpu_kernels.py).
// Straight-line ROF / TV-L1 shapes (prox_g square or abs with scalar a = 1, d = e = 0, b scalar or per pixel; prox_f*
// ind_leq0 with scalar a = 1, d = e = 0), fp32 (4 rows per lane) and fp64 (2 rows per lane), 1 row per lane at other heights; the intermediate iterate is stored nowhere,
// the residual sums of the second iteration are available (RES): BackendPDHG pairs iterations k, k+1 unless k or k+2 is a
// residual iteration.
#include "fused_common.hpp"
#include "reduce.hpp"
#include <type_traits>

namespace prost_hip {

template <class T>
struct IterParamsMc {          // step sizes of one iteration + the host-evaluated step c tau and divisor 1 + step of Function1DSquare
  T tau, sigma, theta;
  T step;
  UniformDiv sq;
  EdgeTerms<T> ec[2];          // GB = 3 (position-dependent Tau): the terms of the classes Tcls[0] (corner), Tcls[1] (edge)
};

template <class T, int VEC, int GB>
struct ColMcX2 {
  T y1[VEC], y2[VEC], x[VEC], b[GB ? VEC : 1];
};

template <class T, int VEC>
__device__ __forceinline__ void ldm_o(const T* __restrict__ base, unsigned byte_off, T (&v)[VEC]) {      // wave-uniform base + lane offset (SADDR form)
  ldv<T, VEC>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off), v);
}
template <class T, int VEC>
__device__ __forceinline__ void stm_o(T* __restrict__ base, unsigned byte_off, const T (&v)[VEC]) {
  stv_nt<T, VEC>(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off), v);
}

// RES: additionally the four residual sums of the SECOND iteration (backend_pdhg.cu:73-120), term by term the expressions of
// fused_iter2d_mc_kernel; K^T y^k of a column waits two steps between stages A and C in LDS, the sums are accumulated in LDS
// (own lanes, no synchronisation) -- as in kernels_fused_iter3d_x2.hip.  One partial (4 doubles) per workgroup.
// GB: 0 scalar b of prox_g, 1 per-pixel b, 2 per-pixel b that carries the mask sentinel (prost_hip_mask_merge: binary a folded in),
// 3 per-pixel b AND the position-dependent primal preconditioner of a gradient handed over as a sparse matrix (FusedArgs::varT; square
// data term) -- the expressions of fused_iter2d_mc_kernel<..., VART>: K^T y in the order of the matrix's transposed CSR row, the pixels of
// the first / last row and column with the step and the divisor of their class
// FMAD: the tolerance-class arithmetic (prost_hip_fused_desc.arith = PROST_HIP_ARITH_FMAD; fp32, GB <= 2): see kernels_fused_iter2.hip
template <class T, int VEC, int GFN, int GB, int LW, bool RES, bool FMAD>
__global__ void __launch_bounds__(kWave * LW, 3) fused_iter2d_mc_x2_kernel(T* __restrict__ x_out, T* __restrict__ y_out, const T* __restrict__ x,
                                                                                const T* __restrict__ y, FusedArgs<T> a, IterParamsMc<T> p1,
                                                                                IterParamsMc<T> p2, double* __restrict__ partial,
                                                                                const PdhgRecord<T>* __restrict__ rec) {
  if (rec) {                       // device-resident step sizes (fused_common.hpp: PdhgRecord): wave-uniform scalar loads
    if (rec->stop) return;
    p1.tau = rec->p.tau; p1.sigma = rec->p.sigma; p1.theta = rec->p.theta; p1.step = rec->p.ug.step; p1.sq = rec->p.ug.sq;
    if (GB == 3) { p1.ec[0] = rec->p.ec[0]; p1.ec[1] = rec->p.ec[1]; }
    p2 = p1;                       // a rule evaluation never falls between the two iterations of a launch
  }
  // two iterations need two valid rows beyond the owned ones on either side: one halo lane of >= 2 rows, or two halo lanes of one row
  // (heights that are not a multiple of 16 bytes of rows: 1 row per lane)
  constexpr int kHalo = VEC >= 2 ? 1 : 2;
  constexpr int kRowsPerWave = (kWave - 2 * kHalo) * VEC;
  constexpr int kPix = kWave * VEC;
  __shared__ T s_sq[2][2][2 * LW][kPix];               // [buffer][stage B / D][component][pixel]
  __shared__ T s_kt[RES ? 3 : 1][RES ? LW : 1][RES ? kPix : 1];
  __shared__ double s_acc[RES ? 4 : 1][RES ? LW : 1][RES ? kWave : 1];
  // column / row indices: 32-bit in the plain instances (wave-uniform orderings are then scalar instructions; 64-bit ones are vector
  // compares), 64-bit in the residual instances, which need 6-8 registers more with 32-bit indices (two of them spill)
  typedef typename std::conditional<RES, long, int>::type idx_t;
  const idx_t nx = (idx_t)a.nx, ny = (idx_t)a.ny;
  const int lane = threadIdx.x & (kWave - 1);
  const int ch = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));      // wave-uniform: base pointers live in SGPRs
  const unsigned total = gridDim.x, chunks = a.chunks;
  const unsigned xcd = blockIdx.x % 8u, q = blockIdx.x / 8u;            // XCD-aware tile order, see kernels_fused_iter.hip
  const unsigned tile = xcd * (total / 8u) + (xcd < total % 8u ? xcd : total % 8u) + q;
  const unsigned strip = tile / chunks, chunk = tile % chunks;
  const idx_t row0 = (idx_t)strip * kRowsPerWave + ((idx_t)lane - kHalo) * VEC;
  const bool active = row0 >= 0 && row0 < ny;
  const bool owner = active && lane >= kHalo && lane < kWave - kHalo;
  const idx_t xa = (idx_t)chunk * a.cols_per_block;
  const idx_t xb = xa + a.cols_per_block < nx ? xa + a.cols_per_block : nx;
  const size_t P = (size_t)nx * (size_t)ny, N = P * LW, plane = (size_t)ch * P;
  const bool tiny_is_zero = a.f_val[1] >= (T)kTinyIsZeroRadius;     // device_math.hpp: norm2_leq0_fast
  const T* y1p = y + plane; const T* y2p = y + N + plane;
  const T* xp = x + plane;
  const T* bp = GB ? a.g_ptr[1] + plane : nullptr;
  const unsigned voff = (unsigned)(row0 * (idx_t)sizeof(T));          // meaningful in active lanes only

  typedef ColMcX2<T, VEC, GB> Col;
  // kUncond (the plain instance): loads are UNCONDITIONAL -- a lane outside the image reads the strip's first rows, a column outside
  // the chunk's range the nearest valid one; nobody uses those values (every use is guarded by the predicate that would have guarded
  // the load), but a register set that is overwritten as a whole each step needs no re-zeroing and no branch around the loads
  // (kernels_fused_iter3d_x2.hip: 10 % there; here, with three wavefronts per workgroup, within the noise -- as is an LDS-only
  // barrier, "s_waitcnt lgkmcnt(0); s_barrier", in place of __syncthreads(): 0.205-0.210 ms per iteration at 4096^2 RGB either way)
  constexpr bool kUncond = !RES;
  const unsigned voff_ld = kUncond && !active ? 0u : voff;
  auto has_col = [&](idx_t k) { return k >= 0 && k < nx && k <= xb + 1; };
  auto load_col = [&](idx_t c, Col& in) {
    if (!kUncond) {
      in = Col{};
      if (!(active && has_col(c))) return;
    }
    const idx_t cc = !kUncond ? c : (c < 0 ? 0 : (c >= nx ? nx - 1 : c));
    const size_t o = (size_t)cc * (size_t)ny;                         // wave-uniform
    ldm_o<T, VEC>(y1p + o, voff_ld, in.y1); ldm_o<T, VEC>(y2p + o, voff_ld, in.y2); ldm_o<T, VEC>(xp + o, voff_ld, in.x);
    if constexpr (GB != 0) ldm_o<T, VEC>(bp + o, voff_ld, in.b);
  };
  // primal step of this channel at column c (backend_pdhg.cu:317-338 with block_gradient2d.cu:122-138 on a zero-filled result);
  // v1 / v2: the dual variable at column c, p1c: its first component at column c-1
  auto primal = [&](idx_t c, const T (&v1)[VEC], const T (&v2)[VEC], const T (&p1c)[VEC], const T (&xin)[VEC], const T (&bv)[GB ? VEC : 1],
                    const IterParamsMc<T>& Pm, T (&xn)[VEC], T (&ktv)[VEC]) {
    constexpr bool VART = GB == 3;
    const T tauT = Pm.tau * a.Tval;
    const T up = lane_up(v2[VEC - 1]);                 // lane 0: no source, its first row is halo
    if constexpr (FMAD) {
      const T rD = (T)Pm.sq.rD;
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const idx_t row = row0 + j;
        const T divy = ((row < ny - 1) ? v2[j] : (T)0) - ((row > 0) ? (j > 0 ? v2[j > 0 ? j - 1 : 0] : up) : (T)0);
        const T divx = ((c < nx - 1) ? v1[j] : (T)0) - ((c > 0) ? p1c[j] : (T)0);
        const T sdiv = divx + divy;
        ktv[j] = -sdiv;
        const T arg = t_fma(tauT, sdiv, xin[j]);
        const T bj = GB ? bv[GB ? j : 0] : a.g_val[1];
        T r;
        if (GFN == PROST_FN_SQUARE) r = t_fma(arg - bj, rD, bj);
        else { const T v = arg - bj; r = (v - t_max(t_min(v, Pm.step), -Pm.step)) + bj; }
        if (GB == 2) r = is_mask_sentinel(bj) ? arg : r;
        xn[j] = r;
      }
      return;
    }
    T parg[VEC], parg0[GB == 2 ? VEC : 1];
    T argv[VART ? VEC : 1];
    bool edgev[VART ? VEC : 1], cornerv[VART ? VEC : 1];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const idx_t row = row0 + j;
      T divy = (row < ny - 1) ? v2[j] : (T)0;
      if (row > 0) divy -= (j > 0 ? v2[j > 0 ? j - 1 : 0] : up);
      T divx = (c < nx - 1) ? v1[j] : (T)0;
      if (c > 0) divx -= p1c[j];
      T kty = (T)0 - (divx + divy);
      T tT = tauT;
      if (VART) {
        T s = 0;                     // K^T y in the order of the matrix's transposed CSR row (kernels_fused_iter.hip: fused_iter2d_kernel)
        if (c > 0) s += p1c[j];
        if (c < nx - 1) s -= v1[j];
        if (row > 0) s += (j > 0 ? v2[j > 0 ? j - 1 : 0] : up);
        if (row < ny - 1) s -= v2[j];
        kty = s;
        const int cnt = 4 - (c == 0 ? 1 : 0) - (c == nx - 1 ? 1 : 0) - (row == 0 ? 1 : 0) - (row == ny - 1 ? 1 : 0);
        edgev[VART ? j : 0] = cnt != 4; cornerv[VART ? j : 0] = cnt == 2;
        tT = Pm.tau * (cnt == 4 ? a.Tval : (cnt == 3 ? a.Tcls[1] : a.Tcls[0]));
      }
      ktv[j] = kty;
      const T arg = xin[j] - tT * kty;
      if (VART) argv[VART ? j : 0] = arg;
      if (GB == 2) parg0[GB == 2 ? j : 0] = arg;
      parg[j] = arg - (GB ? bv[GB ? j : 0] : a.g_val[1]);
    }
    T r[VEC];
    if (GFN == PROST_FN_SQUARE) div_to_float_exact_vec<VEC>(parg, Pm.sq, r);
    else {
#pragma unroll
      for (int j = 0; j < VEC; j++) r[j] = f1d_apply<T, GFN>(a.g_fn, parg[j], Pm.step, a.g_val[5], a.g_val[6]);
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) xn[j] = r[j] + (GB ? bv[GB ? j : 0] : a.g_val[1]);
    if (VART) {                      // pixels with their own Tau_j: the divisor 1 + step_j of their class (IterParamsMc::ec)
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        if (edgev[VART ? j : 0]) {
          const bool cn = cornerv[VART ? j : 0];
          UniformDiv dv;
          dv.D = cn ? Pm.ec[0].sq.D : Pm.ec[1].sq.D; dv.rD = cn ? Pm.ec[0].sq.rD : Pm.ec[1].sq.rD;
          const T bj = bv[GB ? j : 0];
          xn[j] = div_to_float_exact(argv[VART ? j : 0] - bj, dv) + bj;
        }
      }
    }
    if (GB == 2) {
      // merged b stream: where the binary coefficient a of prox_g is 0 the element passes through (elem_operation_1d.hpp:42-44 with d = e = 0)
#pragma unroll
      for (int j = 0; j < VEC; j++) if (is_mask_sentinel(bv[GB ? j : 0])) xn[j] = parg0[GB == 2 ? j : 0];
    }
  };
  // first half of the dual step at column c (backend_pdhg.cu:341-370, block_gradient2d.cu:61-77): the two dual arguments of this
  // channel, their squares published for the norm over all channels
  auto dual_args = [&](idx_t c, const T (&xn_c)[VEC], const T (&xn_n)[VEC], const T (&xo_c)[VEC], const T (&xo_n)[VEC], const T (&v1)[VEC],
                       const T (&v2)[VEC], const IterParamsMc<T>& Pm, T (&av)[2][VEC], T (&sq)[2 * LW][kPix]) {
    const T sigS = Pm.sigma * a.Sval, theta = Pm.theta;
    const bool has_next = c + 1 < nx;
    const T bel_n = lane_down(xn_c[0]);                // lane 63: no source, its last row is halo
    const T bel_o = lane_down(xo_c[0]);
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const idx_t row = row0 + j;
      const T below_n = (j < VEC - 1) ? xn_c[j < VEC - 1 ? j + 1 : 0] : bel_n;
      const T below_o = (j < VEC - 1) ? xo_c[j < VEC - 1 ? j + 1 : 0] : bel_o;
      const T kx0 = has_next ? xn_n[j] - xn_c[j] : (T)0;
      const T kx1 = (row < ny - 1) ? below_n - xn_c[j] : (T)0;
      const T kp0 = has_next ? xo_n[j] - xo_c[j] : (T)0;
      const T kp1 = (row < ny - 1) ? below_o - xo_c[j] : (T)0;
      if constexpr (FMAD) {
        av[0][j] = t_fma(sigS, t_fma(1 + theta, kx0, -(theta * kp0)), v1[j]);
        av[1][j] = t_fma(sigS, t_fma(1 + theta, kx1, -(theta * kp1)), v2[j]);
      } else {
      av[0][j] = v1[j] + sigS * ((1 + theta) * kx0 - theta * kp0);              // backend_pdhg.cu:54-70
      av[1][j] = v2[j] + sigS * ((1 + theta) * kx1 - theta * kp1);
      }
      sq[ch][j * kWave + lane] = av[0][j] * av[0][j];                           // [j][lane]: conflict-free banks
      sq[LW + ch][j * kWave + lane] = av[1][j] * av[1][j];
    }
  };
  // second half, after the barrier: the norm in the component order of ElemOperationNorm2 (d/dx of all channels, then d/dy), the projection
  auto dual_finish = [&](const T (&av)[2][VEC], const T (&sq)[2 * LW][kPix], T (&out)[2][VEC]) {
    T nv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      T norm = 0;
#pragma unroll
      for (int i = 0; i < 2 * LW; i++) norm += sq[i][j * kWave + lane];
      nv[j] = norm;
      if constexpr (FMAD) {          // pr v / ||v|| = v min(b / ||v||, 1), radius b > 0 (host-checked); ||v|| = 0: b * inf -> factor 1 on a zero vector
        const T sc = t_min(a.f_val[1] * t_rsq(norm), (T)1);
        out[0][j] = av[0][j] * sc; out[1][j] = av[1][j] * sc;
      }
    }
    if constexpr (!FMAD) norm2_leq0_fast<T, 2, VEC>(nv, av, a.f_val[1], tiny_is_zero, out);
  };

  // residual sums of the second iteration (RES): the terms of fused_iter2d_mc_kernel
  const T sqT = t_sqrt(a.Tval), sqS = t_sqrt(a.Sval);
  const SharedDivisor<T> div_tauT(p2.tau * sqT), div_sigS(p2.sigma * sqS);   // wave-uniform: exact quotients through one double reciprocal each
  if (RES) {
#pragma unroll
    for (int k = 0; k < 4; k++) s_acc[RES ? k : 0][RES ? ch : 0][RES ? lane : 0] = 0;   // primal diff^2, primal var^2, dual diff^2, dual var^2
  }
  auto accumulate = [&](int k, double v) { s_acc[RES ? k : 0][RES ? ch : 0][RES ? lane : 0] += v; };
  // dual_residual_transform (backend_pdhg.cu:73-94) at the column of stage C: xo / xn = x^(k+1) / x^(k+2), kt_prev = K^T y^k, kt = K^T y^(k+1)
  auto dual_residual = [&](idx_t c, const T (&xo)[VEC], const T (&xn)[VEC], const T (&kt_prev)[VEC], const T (&kt)[VEC]) {
    double dd = 0, dv = 0;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      T sT = sqT;
      bool edge = false;
      if (GB == 3) {                 // sqrt(Tau_j) of this pixel's class
        const idx_t row = row0 + j;
        const int cnt = 4 - (c == 0 ? 1 : 0) - (c == nx - 1 ? 1 : 0) - (row == 0 ? 1 : 0) - (row == ny - 1 ? 1 : 0);
        edge = cnt != 4;
        if (edge) sT = t_sqrt(cnt == 3 ? a.Tcls[1] : a.Tcls[0]);
      }
      const T w_hat = edge ? (xo[j] - xn[j]) / (p2.tau * sT) - sT * kt_prev[j] : div_tauT.div(xo[j] - xn[j]) - sqT * kt_prev[j];
      const T diff = w_hat + sT * kt[j];
      dd += (double)(diff * diff); dv += (double)(w_hat * w_hat);
    }
    if (owner) { accumulate(2, dd); accumulate(3, dv); }
  };
  // primal_residual_transform (backend_pdhg.cu:97-120) at the column of stage D: v* = y^(k+1), out = y^(k+2); K x, K x_prev formed again
  auto primal_residual = [&](idx_t c, const T (&xn_c)[VEC], const T (&xn_n)[VEC], const T (&xo_c)[VEC], const T (&xo_n)[VEC], const T (&v1)[VEC],
                             const T (&v2)[VEC], const T (&out)[2][VEC]) {
    const T theta = p2.theta;
    const bool has_next = c + 1 < nx;
    const T bel_n = lane_down(xn_c[0]);
    const T bel_o = lane_down(xo_c[0]);
    double pd = 0, pv = 0;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const idx_t row = row0 + j;
      const T below_n = (j < VEC - 1) ? xn_c[j < VEC - 1 ? j + 1 : 0] : bel_n;
      const T below_o = (j < VEC - 1) ? xo_c[j < VEC - 1 ? j + 1 : 0] : bel_o;
      const T kx[2] = {has_next ? xn_n[j] - xn_c[j] : (T)0, (row < ny - 1) ? below_n - xn_c[j] : (T)0};
      const T kp[2] = {has_next ? xo_n[j] - xo_c[j] : (T)0, (row < ny - 1) ? below_o - xo_c[j] : (T)0};
      const T yv[2] = {v1[j], v2[j]};
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const T z_hat = div_sigS.div(yv[i] - out[i][j]) + sqS * ((1 + theta) * kx[i] - theta * kp[i]);
        const T diff = z_hat - sqS * kx[i];
        pd += (double)(diff * diff); pv += (double)(z_hat * z_hat);
      }
    }
    if (owner) { accumulate(0, pd); accumulate(1, pv); }
  };

  Col in1 = {}, in2 = {};                              // raw columns c+1 and c+2
  T b_c[GB ? VEC : 1];                                 // b of prox_g at column c (stage C)
  T x1_m[VEC], x1_0[VEC], x1_1[VEC], x1_2[VEC];        // x^(k+1) at columns c-1 .. c+2
  T ya_m[VEC], yb_m[VEC], ya_0[VEC], yb_0[VEC], ya_1[VEC], yb_1[VEC];   // y^(k+1) at columns c-1, c, c+1
  T x2_m[VEC], x2_0[VEC];                              // x^(k+2) at columns c-1, c
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    x1_m[j] = x1_0[j] = x1_1[j] = x1_2[j] = 0;
    ya_m[j] = yb_m[j] = ya_0[j] = yb_0[j] = ya_1[j] = yb_1[j] = 0;
    x2_m[j] = x2_0[j] = 0;
  }
#pragma unroll
  for (int j = 0; j < (GB ? VEC : 1); j++) b_c[j] = 0;
  if (active && xa - 2 >= 0) ldm_o<T, VEC>(y1p + (size_t)(xa - 2) * (size_t)ny, voff, in2.y1);   // becomes in1.y1 for A(xa-1)
  Col pre = {};
  load_col(xa - 1, pre);
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): nothing of the prologue in flight when the loop starts (see kernels_fused_iter3d_x2.hip)

  int k3 = 0;                                          // slot of s_kt that stage A writes in this step
  for (idx_t c = xa - 3; c <= xb; c++) {
    const int buf = (int)((c + 4) & 1);
    in1 = in2; in2 = pre;
    load_col(c + 3, pre);
    const idx_t ca = c + 2, cb = c + 1, cd = c - 1;
    const bool runA = ca >= (xa - 1 > 0 ? xa - 1 : 0) && ca < nx && ca <= xb + 1;
    const bool runB = cb >= (xa - 1 > 0 ? xa - 1 : 0) && cb < nx && cb <= xb;
    const bool runC = c >= xa && c < nx && c <= xb;
    const bool runD = cd >= xa && cd < xb;
    T avB[2][VEC], avD[2][VEC];
    if (runA) {                                                                                    // stage A
      T kt_a[VEC];
      primal(ca, in2.y1, in2.y2, in1.y1, in2.x, in2.b, p1, x1_2, kt_a);
      if (RES) {
#pragma unroll
        for (int j = 0; j < VEC; j++) s_kt[RES ? k3 : 0][RES ? ch : 0][RES ? j * kWave + lane : 0] = kt_a[j];     // K^T y^k at column c+2: read by stage C two steps later
      }
    }
    if (runB) dual_args(cb, x1_1, x1_2, in1.x, in2.x, in1.y1, in1.y2, p1, avB, s_sq[buf][0]);      // stage B, first half
    if (runC) {                                                                                    // stage C
      T kt_c[VEC];
      primal(c, ya_0, yb_0, ya_m, x1_0, b_c, p2, x2_0, kt_c);
      if (RES && c < xb && (size_t)c >= a.rx0 && (size_t)c < a.rx1) {
        T kt_0[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) kt_0[j] = s_kt[RES ? (k3 + 1) % 3 : 0][RES ? ch : 0][RES ? j * kWave + lane : 0];     // written two steps ago
        dual_residual(c, x1_0, x2_0, kt_0, kt_c);
      }
    }
    if (runD) dual_args(cd, x2_m, x2_0, x1_m, x1_0, ya_m, yb_m, p2, avD, s_sq[buf][1]);            // stage D, first half
    __syncthreads();                                   // one barrier per column step (every wavefront runs the same stages)
    if (runB) {
      T o[2][VEC];
      dual_finish(avB, s_sq[buf][0], o);
#pragma unroll
      for (int j = 0; j < VEC; j++) { ya_1[j] = o[0][j]; yb_1[j] = o[1][j]; }
    }
    if (runD) {
      T o[2][VEC];
      dual_finish(avD, s_sq[buf][1], o);
      if (RES && (size_t)cd >= a.rx0 && (size_t)cd < a.rx1) primal_residual(cd, x2_m, x2_0, x1_m, x1_0, ya_m, yb_m, o);
      if (owner) {
        const size_t off = plane + (size_t)cd * (size_t)ny;           // wave-uniform
        stm_o<T, VEC>(y_out + off, voff, o[0]); stm_o<T, VEC>(y_out + N + off, voff, o[1]);
      }
    }
    if (runC && owner && c < xb) stm_o<T, VEC>(x_out + plane + (size_t)c * (size_t)ny, voff, x2_0);
    // shift the pipeline by one column
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      x1_m[j] = x1_0[j]; x1_0[j] = x1_1[j]; x1_1[j] = x1_2[j];
      ya_m[j] = ya_0[j]; yb_m[j] = yb_0[j];
      ya_0[j] = ya_1[j]; yb_0[j] = yb_1[j];
      x2_m[j] = x2_0[j];
    }
#pragma unroll
    for (int j = 0; j < (GB ? VEC : 1); j++) b_c[j] = in1.b[j];
    k3 = k3 == 2 ? 0 : k3 + 1;
  }
  if (RES) {
    __syncthreads();                                   // every wavefront has finished reading the squares: the buffer is free
    double* sred = reinterpret_cast<double*>(&s_sq[0][0][0][0]);
    double r_pd = s_acc[0][RES ? ch : 0][RES ? lane : 0], r_pv = s_acc[RES ? 1 : 0][RES ? ch : 0][RES ? lane : 0], r_dd = s_acc[RES ? 2 : 0][RES ? ch : 0][RES ? lane : 0],
           r_dv = s_acc[RES ? 3 : 0][RES ? ch : 0][RES ? lane : 0];
    r_pd = wave_sum(r_pd); r_pv = wave_sum(r_pv); r_dd = wave_sum(r_dd); r_dv = wave_sum(r_dv);
    if (lane == 0) { sred[4 * ch + 0] = r_pd; sred[4 * ch + 1] = r_pv; sred[4 * ch + 2] = r_dd; sred[4 * ch + 3] = r_dv; }
    __syncthreads();
    if (threadIdx.x < 4) {
      double t = 0;
      for (int w = 0; w < LW; w++) t += sred[4 * w + threadIdx.x];       // fixed order: run-to-run deterministic
      partial[4 * (size_t)blockIdx.x + threadIdx.x] = t;
    }
  }
}

// rows per lane: 16 bytes where the height allows, otherwise 1; owned rows of a wavefront (one halo lane of >= 2 rows or two of 1 row on either side)
static int mc_x2_vec(int dtype, size_t ny) { const int full = dtype == 0 ? 4 : 2; return ny % (size_t)full == 0 ? full : 1; }
static size_t mc_x2_rows(int dtype, size_t ny) { const int v = mc_x2_vec(dtype, ny); return v >= 2 ? (size_t)(kWave - 2) * v : (size_t)(kWave - 4); }

static bool iter_mc_x2_ok(const prost_hip_fused_desc* d, int dtype) {
  if ((dtype != 0 && dtype != 1) || !d || d->is3d || d->f_moreau || d->L < 2 || d->L > 4) return false;
  if (d->var_T && (d->g_fn != PROST_FN_SQUARE || !d->g_coeff_ptr[1] || d->g_b_masked)) return false;   // position-dependent Tau: square data term with per-pixel b
  if (d->nx < 4 || d->ny < 4) return false;
  if ((d->g_fn != PROST_FN_SQUARE && d->g_fn != PROST_FN_ABS) || d->f_fn != PROST_FN_IND_LEQ0) return false;
  for (int k = 0; k < 7; k++) {
    if (d->f_coeff_ptr[k]) return false;
    if (k != 1 && d->g_coeff_ptr[k]) return false;
  }
  if (d->g_coeff_ptr[1] && !aligned16(d->g_coeff_ptr[1])) return false;
  if (d->g_b_masked && (!d->g_coeff_ptr[1] || d->g_fn != PROST_FN_SQUARE)) return false;
  if (d->g_coeff_val[0] != 1.0 || d->g_coeff_val[2] == 0.0 || d->g_coeff_val[3] != 0.0 || d->g_coeff_val[4] != 0.0) return false;
  if (d->f_coeff_val[0] != 1.0 || d->f_coeff_val[3] != 0.0 || d->f_coeff_val[4] != 0.0) return false;
  // (res_x0 / res_x1 -- the owned columns of a sharded slab -- restrict the residual sums of the RES instances)
  if ((double)d->nx * (double)d->ny * (dtype == 0 ? 4.0 : 8.0) >= 4294967296.0) return false;              // 32-bit byte offsets per plane
  const size_t strips = (d->ny + mc_x2_rows(dtype, d->ny) - 1) / mc_x2_rows(dtype, d->ny);
  if (strips > (size_t)kReduceBlocks / 2) return false;              // residual launches: one partial per workgroup, at best one chunk per strip
  return strips * d->nx < (size_t)1 << 31;
}

// chunk length: a launch lasts about (rounds of wavefronts on the 3 resident slots per SIMD) x (column steps of a wavefront), so
// the chunk length minimises ceil(strips * chunks * L / slots) * (columns + 3 warm-up steps): long chunks once the image fills
// the chip, short ones for small images, where every wavefront gets a slot anyway and only
// the number of steps counts (512^2 RGB: 2 columns = 5 steps per two iterations)
static size_t mc_x2_chunk_cols(const prost_hip_fused_desc* d, int dtype, int cols, bool res) {
  if (cols > 0) return (size_t)cols < d->nx ? (size_t)cols : d->nx;
  const size_t strips = (d->ny + mc_x2_rows(dtype, d->ny) - 1) / mc_x2_rows(dtype, d->ny);
  const size_t slots = 256 * 4 * 3;
  size_t best_c = 1, best_cost = (size_t)-1;
  // (capped at 24 columns: beyond the point where every slot is taken, MORE and shorter workgroups hide the per-column barrier
  // better than fewer warm-up columns pay -- 4096^2 RGB: 24 columns 0.186 ms per iteration, 72 columns 0.198, 96 columns 0.220)
  // residual launches: one partial (4 doubles) per workgroup -- where 24 columns leave more workgroups than the workspace holds
  // (4096^2 in fp64: 34 strips of 124 rows), the cap moves to the shortest chunk that fits (iter_mc_x2_ok: one chunk always does)
  const size_t max_groups = (size_t)kReduceBlocks / 2;
  size_t cap = 24;
  if (res && strips * ((d->nx + cap - 1) / cap) > max_groups) cap = (d->nx + max_groups / strips - 1) / (max_groups / strips);
  for (size_t c = 1; c <= cap && c <= d->nx; c++) {
    const size_t waves = strips * ((d->nx + c - 1) / c) * d->L;
    if (res && strips * ((d->nx + c - 1) / c) > max_groups) continue;
    const size_t cost = ((waves + slots - 1) / slots) * (c + 3);
    if (cost <= best_cost) { best_cost = cost; best_c = c; }         // ties: the longer chunk (less redundant arithmetic)
  }
  return best_c < d->nx ? best_c : d->nx;
}

// tolerance-class instances: fp32, 4 rows per lane (heights that are a multiple of 4), uniform Tau, ind_leq0 radius > 0; anything else computes exactly
static bool iter_mc_x2_fmad(const prost_hip_fused_desc* d, int dtype) {
  return d->arith == PROST_HIP_ARITH_FMAD && dtype == 0 && iter_mc_x2_ok(d, dtype) && mc_x2_vec(dtype, d->ny) == 4 && !d->var_T && d->f_coeff_val[1] > 0.0;
}

template <class T>
static int run_iter_mc_x2(const prost_hip_fused_desc* d, T* x_out, T* y_out, const T* x, const T* y, const double* tau, const double* sigma,
                          const double* theta, int cols, double* out4, void* ws, void* stream, void* record = nullptr, const RuleTail* tail = nullptr) {
  const PdhgRecord<T>* rec = static_cast<const PdhgRecord<T>*>(record);
  const double one2[2] = {1.0, 1.0};
  if (rec) { tau = sigma = theta = one2; }          // (the shape checks below do not depend on the step sizes: e = 0 is part of the supported shape)
  constexpr int kDtype = std::is_same<T, float>::value ? 0 : 1;
  if (!iter_mc_x2_ok(d, kDtype)) { set_error("fused multi-channel double iteration: unsupported description (see prost_hip_fused_iteration_mc_x2_supported)"); return 1; }
  if (!aligned16(x_out) || !aligned16(y_out) || !aligned16(x) || !aligned16(y)) { set_error("fused multi-channel double iteration: vectors must be 16-byte aligned"); return 1; }
  if (x_out == x || y_out == y) { set_error("fused multi-channel double iteration: outputs must not alias inputs"); return 1; }
  if (out4 && !ws) { set_error("fused multi-channel double iteration: residuals need the reduction workspace"); return 1; }
  FusedArgs<T> a = make_fused_args<T>(d);
  IterParamsMc<T> p[2];
  for (int i = 0; i < 2; i++) {
    p[i].tau = (T)tau[i]; p[i].sigma = (T)sigma[i]; p[i].theta = (T)theta[i];
    const UniformProx<T> ug = make_uniform_prox<T>(a.g_val, (T)tau[i] * a.Tval);
    const UniformProx<T> uf = make_uniform_prox<T>(a.f_val, (T)sigma[i] * a.Sval);
    if (!ug.a_one || !ug.den_one || ug.degenerate || !uf.a_one || !uf.den_one) { set_error("fused multi-channel double iteration: not the straight-line ROF / TV-L1 shape"); return 1; }
    p[i].sq = ug.sq; p[i].step = ug.step;
    for (int k = 0; k < 2; k++) p[i].ec[k] = a.varT ? make_edge_terms<T>(a.g_val, (T)tau[i] * a.Tcls[k]) : EdgeTerms<T>();
  }
  const size_t rows = mc_x2_rows(kDtype, d->ny);
  const size_t strips = (d->ny + rows - 1) / rows;
  const size_t c = mc_x2_chunk_cols(d, kDtype, cols, out4 != nullptr);
  a.cols_per_block = (int)c;
  a.chunks = (unsigned)((d->nx + c - 1) / c);
  const unsigned grid = (unsigned)(strips * a.chunks);
  hipStream_t s = as_stream(stream);
  if (out4 && grid > (unsigned)kReduceBlocks / 2) { set_error("fused multi-channel double iteration: grid exceeds the reduction workspace"); return 1; }
  double* partial = static_cast<double*>(ws);
  const bool full = mc_x2_vec(kDtype, d->ny) > 1;
  const bool fmad = iter_mc_x2_fmad(d, kDtype);
#define GO6(VV, G, B, LWv, R, A) PH_LAUNCH((fused_iter2d_mc_x2_kernel<T, VV, G, B, LWv, R, A>), dim3(grid), dim3(kWave * LWv), 0, s, x_out, y_out, x, y, a, p[0], p[1], partial, rec)
#define GO5(VV, G, B, LWv, R) do { bool done_ = false; if constexpr (std::is_same<T, float>::value && VV == 4 && B != 3) { if (fmad) { GO6(VV, G, B, LWv, \
    R, true); done_ = true; } } if (!done_) GO6(VV, G, B, LWv, R, false); } while (0)
#define GO4(G, B, LWv, R) do { if (full) GO5(VecOf<T>::N, G, B, LWv, R); else GO5(1, G, B, LWv, R); } while (0)
#define GO3(G, B, LWv) do { if (out4) GO4(G, B, LWv, true); else GO4(G, B, LWv, false); } while (0)
#define GO2(G, B) do { if (d->L == 2) GO3(G, B, 2); else if (d->L == 3) GO3(G, B, 3); else GO3(G, B, 4); } while (0)
#define GO(B) do { if (d->g_fn == PROST_FN_ABS) GO2(PROST_FN_ABS, B); else GO2(PROST_FN_SQUARE, B); } while (0)
  if (a.varT) GO2(PROST_FN_SQUARE, 3);                                    // the gradient handed over as a sparse matrix (iter_mc_x2_ok: square, per-pixel b)
  else if (d->g_coeff_ptr[1] && d->g_b_masked) GO2(PROST_FN_SQUARE, 2);  // inpainting: binary mask folded into b (square data term)
  else if (d->g_coeff_ptr[1]) GO(1); else GO(0);
#undef GO
#undef GO2
#undef GO3
#undef GO4
#undef GO5
#undef GO6
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "fused multi-channel double iteration kernel"); }
  if (out4 && tail && tail->apply) return launch_fold4_rule<T>(out4, partial, grid, record, tail->iteration, tail->mirror, s);
  if (out4) return launch_fold4(out4, partial, grid, s);
  return 0;
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_fused_iteration_mc_x2_supported(const prost_hip_fused_desc* d, int dtype) { return iter_mc_x2_ok(d, dtype) ? 1 : 0; }
// 1 iff the launch is also FASTER than two single launches.  Measured per iteration, RGB: 256^2 6.4 vs 8.5 us (launch-bound: one
// launch instead of two), 384^2 8.0 vs 7.3, 512^2 10.6 vs 10.5, 700 x 464 10.5 vs 10.0 (latency-bound: a column step of the pair
// pipeline issues twice the instructions of a single step and there are too few wavefronts to hide it), 768^2 14.8 vs 15.2,
// 1024^2 21.9 vs 25.3, 2048^2 56 vs 87, 4096^2 187 vs 301 (throughput-bound: half the HBM traffic).
int prost_hip_fused_iteration_mc_x2_profitable(const prost_hip_fused_desc* d, int dtype) {
  if (!iter_mc_x2_ok(d, dtype)) return 0;
  const double values = (double)d->nx * (double)d->ny * (double)d->L;
  return values <= 3.0e5 || values >= 1.5e6 ? 1 : 0;
}
int prost_hip_fused_iteration_mc_x2_arith(const prost_hip_fused_desc* d, int dtype) { return iter_mc_x2_fmad(d, dtype) ? PROST_HIP_ARITH_FMAD : PROST_HIP_ARITH_EXACT; }
int prost_hip_fused_iteration_mc_x2_chunk_cols(const prost_hip_fused_desc* d, int dtype, int with_residuals) {
  return iter_mc_x2_ok(d, dtype) ? (int)mc_x2_chunk_cols(d, dtype, 0, with_residuals != 0) : 0;
}
int prost_hip_fused_iteration_mc_x2_f32(const prost_hip_fused_desc* d, float* x_out, float* y_out, const float* x, const float* y, const double* tau,
                                        const double* sigma, const double* theta, int cols, double* res_out4, void* workspace, void* stream) {
  return run_iter_mc_x2<float>(d, x_out, y_out, x, y, tau, sigma, theta, cols, res_out4, workspace, stream);
}
int prost_hip_fused_iteration_mc_x2_f64(const prost_hip_fused_desc* d, double* x_out, double* y_out, const double* x, const double* y, const double* tau,
                                        const double* sigma, const double* theta, int cols, double* res_out4, void* workspace, void* stream) {
  return run_iter_mc_x2<double>(d, x_out, y_out, x, y, tau, sigma, theta, cols, res_out4, workspace, stream);
}
int prost_hip_fused_iteration_mc_x2_rec_f32(const prost_hip_fused_desc* d, float* x_out, float* y_out, const float* x, const float* y, void* record, int cols,
                                            double* res_out4, void* workspace, int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!record) { set_error("fused multi-channel double iteration: no step-size record"); return 1; }
  const RuleTail tail = {apply_rule, iteration, mirror};
  return run_iter_mc_x2<float>(d, x_out, y_out, x, y, nullptr, nullptr, nullptr, cols, res_out4, workspace, stream, record, &tail);
}
int prost_hip_fused_iteration_mc_x2_rec_f64(const prost_hip_fused_desc* d, double* x_out, double* y_out, const double* x, const double* y, void* record, int cols,
                                            double* res_out4, void* workspace, int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!record) { set_error("fused multi-channel double iteration: no step-size record"); return 1; }
  const RuleTail tail = {apply_rule, iteration, mirror};
  return run_iter_mc_x2<double>(d, x_out, y_out, x, y, nullptr, nullptr, nullptr, cols, res_out4, workspace, stream, record, &tail);
}
}  // extern "C"
