// kernels_fused_iter3d_x2.hip -- gradient3d, TWO PDHG iterations per kernel launch (temporal blocking in 3-D).
//
// fused_iter3d_pw_kernel (one iteration per launch, planes across the wavefronts of a workgroup) is memory-bound: 2.0-2.7 ms
// between boxes at 2048 x 2048 x 64 against 1.28 ms with every access served from cache.  Here a workgroup of WT = 16
// wavefronts owns P = WT - 3 = 13 consecutive planes of one (row strip, column chunk) and every wavefront runs the 4-stage
// column pipeline of the 2-D pair kernel (kernels_fused_iter2.hip) on ITS plane:
//     A(c+2): x1 = primal step of iteration k            B(c+1): y1 = dual step of iteration k
//     C(c)  : x2 = primal step of iteration k+1          D(c-1): y2 = dual step of iteration k+1
// What a stage needs from a neighbouring plane -- x1(l+1) for B, the third component of y1(l-1) for C, x2(l+1) for D -- was
// published through LDS in the PREVIOUS column step by the wavefront of that plane.  The residual instance orders the exchange with
// ONE workgroup barrier per column (double-buffered LDS); the plain instance (round 3) with per-wavefront progress counters that
// couple NEIGHBOURING planes only (three slots; FLAGS below) and with unconditional loads -- same box, 2048 x 2048 x 64, plain launch:
// 1.497 -> 1.325 ms per iteration.  Planes l0 .. l0+P-1 are owned (x^(k+2), y^(k+2) stored); the wavefront of plane l0-1 runs A
// and B only, that of plane l0+P runs A, B, C, that of plane l0+P+1 runs A only (helper planes: 4P+6 = 58 stage units per 52
// useful ones).  Row neighbours come from adjacent lanes (DPP), lanes 0 and 63 are halo lanes as in the 2-D pair kernel.
//
// Shape of the instance (measured at 2048 x 2048 x 64 fp32, per iteration):
//   4 rows per lane x 8 wavefronts (5 owned planes, 212 VGPRs, 2 waves per SIMD): 2.22 ms -- arithmetic alone 1.69 ms, 30 % of
//   it on helper planes, the barrier 0.39 ms;
//   2 rows per lane x 16 wavefronts (13 owned planes, 111 VGPRs, 4 waves per SIMD): arithmetic alone 1.15 ms, memory alone
//   1.2 ms, together 1.5-1.6 ms.  The HBM traffic of a launch is 11.15 GB (rocprofv3 counters: 6.62 GB read + 4.53 GB written;
//   ideal 5 + 4 floats per voxel = 9.66 GB) for TWO iterations, against 11.3 GB for ONE of the single-iteration kernel.
// Every stage evaluates the expressions of kernels_fused_iter3d_pw.hip / kernels_fused3d.hip, so x^(k+2), y^(k+2) are
// bit-identical to two single launches (tests/test_gpu_fused3d.py).  Straight-line ROF / TV-L1 shapes only (prox_g square or abs with
// scalar a = 1, d = e = 0, b scalar or per voxel; prox_f* ind_leq0 with scalar a = 1, d = e = 0); fp32 with 2 rows per lane (even heights); fp64
// and fp32 at odd heights with 1 row per lane and two halo lanes on either side.  The
// intermediate iterate is stored nowhere; the residual sums of the second iteration are available (RES).
#include "fused_common.hpp"
#include "reduce.hpp"
#include <cstdlib>
#include <type_traits>

namespace prost_hip {

template <class T>
struct IterParams3 {           // step sizes of one iteration + the host-evaluated step c tau and divisor 1 + step of Function1DSquare
  T tau, sigma, theta;
  T step;
  UniformDiv sq;
};

// VEC consecutive rows per lane (4: 16-byte accesses, 2: 8-byte accesses)
template <class T, int VEC>
__device__ __forceinline__ void ldx(const T* __restrict__ p, T (&v)[VEC]) {
  if constexpr (VEC == 1) { v[0] = p[0]; return; }
  typedef T V __attribute__((ext_vector_type(VEC > 1 ? VEC : 2)));
  const V t = *reinterpret_cast<const V*>(p);
#pragma unroll
  for (int j = 0; j < VEC; j++) v[j] = t[j];
}
template <class T, int VEC>
__device__ __forceinline__ void stx_nt(T* __restrict__ p, const T (&v)[VEC]) {
  if constexpr (VEC == 1) { __builtin_nontemporal_store(v[0], p); return; }
  typedef T V __attribute__((ext_vector_type(VEC > 1 ? VEC : 2)));
  V t;
#pragma unroll
  for (int j = 0; j < VEC; j++) t[j] = v[j];
  __builtin_nontemporal_store(t, reinterpret_cast<V*>(p));
}

// wave-uniform base + 32-bit per-lane byte offset: the global_load / global_store SADDR form, no 64-bit address VGPRs
// (128 -> 111 VGPRs)
template <class T, int VEC>
__device__ __forceinline__ void ldx_o(const T* __restrict__ base, unsigned byte_off, T (&v)[VEC]) {
  ldx<T, VEC>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off), v);
}
template <class T, int VEC>
__device__ __forceinline__ void stx_o(T* __restrict__ base, unsigned byte_off, const T (&v)[VEC]) {
  stx_nt<T, VEC>(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off), v);
}

// Progress counters of the FLAGS instance, accessed so that ONLY LDS traffic is ordered.  A workgroup-scope fence (and
// __syncthreads()) also waits for the wavefront's global loads and stores (s_waitcnt vmcnt(0)); the exchange buffers live in LDS and
// a wavefront's LDS operations are processed in order, so lgkmcnt is all that has to be waited for, and the "memory" clobbers keep
// the compiler from moving LDS accesses across these points.  Same box, 2048 x 2048 x 64: 1.325 ms per iteration against 1.355 with
// fences.  (The barrier of the residual instance stays __syncthreads(): with the asm barrier that instance spills 9 registers instead
// of 4 and runs 2.0 instead of 1.72 ms per iteration.)
__device__ __forceinline__ int lds_flag_read(const int* p) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
  return v;
}
__device__ __forceinline__ void lds_flag_write(int* p, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)p), "v"(v) : "memory"); }

template <class T, int VEC, bool GB>
struct ColX2 {
  T y1[VEC], y2[VEC], y3[VEC], x[VEC], b[GB ? VEC : 1];   // own plane
  T zx[VEC];                                              // x of the plane above (old iterate, for K x_prev of stage B)
  T y3m[VEC];                                             // y3 of the plane below (stage A)
};

// RES: additionally the four residual sums of the SECOND iteration (primal_residual_transform / dual_residual_transform,
// backend_pdhg.cu:73-120), term by term the expressions of fused_iter3d_kernel: K^T y^k at column c comes from stage A two
// steps earlier, everything else is in registers when stages C and D run.  One partial (4 doubles) per workgroup.
// FLAGS: the workgroup barrier per column step is replaced by per-wavefront progress counters in LDS.  A wavefront needs, at
// step k, only what the wavefronts of the two NEIGHBOURING planes published at step k - 1; with the exchange buffers three
// slots deep (slot k % 3 is written at step k, slot (k - 1) % 3 read) it may start step k as soon as both neighbours have
// finished step k - 1 -- which also guarantees that they are done reading the slot it overwrites (written at step k - 3, read
// at their step k - 2).  A barrier makes every step as long as its slowest wavefront (16 of them, each waiting for seven
// loads); with the counters a late load delays the neighbouring planes by one step and is absorbed further out.
// FMAD: the tolerance-class arithmetic (prost_hip_fused_desc.arith = PROST_HIP_ARITH_FMAD, fp32): fused multiply-adds where a product feeds a
// sum, the quotient by 1 + step as a product with its fp32 reciprocal, pr v / ||v|| as v min(b rsq(||v||^2), 1) -- see
// kernels_fused_iter2.hip; results within the tolerance of tests/test_gpu_fmad.py of the exact instances, not bit-identical to them.
template <class T, int VEC, int GFN, bool GB, int WT, bool RES, bool FLAGS, bool FMAD>
__global__ void __launch_bounds__(kWave * WT, 1) fused_iter3d_x2_kernel(T* __restrict__ x_out, T* __restrict__ y_out, const T* __restrict__ x,
                                                                       const T* __restrict__ y, FusedArgs<T> a, IterParams3<T> p1, IterParams3<T> p2,
                                                                       double* __restrict__ partial, const PdhgRecord<T>* __restrict__ rec) {
  if (rec) {                       // device-resident step sizes (fused_common.hpp: PdhgRecord): both iterations of the launch run with the record's
    if (rec->stop) return;         // values -- no rule evaluation falls between them (the residual sums, if any, are those of the second)
    p1.tau = rec->p.tau; p1.sigma = rec->p.sigma; p1.theta = rec->p.theta; p1.step = rec->p.ug.step; p1.sq = rec->p.ug.sq;
    p2 = p1;
  }
  // two iterations need TWO valid rows beyond the owned ones on either side (x^(k+2) of a row needs y^(k+1) of the row above, that
  // x^(k+1) of the same row, that the raw y of the row above it): one halo lane of >= 2 rows, or two halo lanes of one row (fp64)
  constexpr int kHalo = VEC >= 2 ? 1 : 2;
  constexpr int kRowsPerWave = (kWave - 2 * kHalo) * VEC;
  constexpr int kPix = kWave * VEC;
  constexpr int P = WT - 3;                            // planes a workgroup owns
  constexpr int kSlots = FLAGS ? 3 : 2;
  __shared__ T s_x1[kSlots][WT][kPix], s_y3[kSlots][WT][kPix], s_x2[kSlots][WT][kPix];
  __shared__ int s_step[FLAGS ? WT : 1];               // FLAGS: last step whose publishes wavefront w has finished
  // RES: K^T y^k of the own plane waits two steps between stages A and C, and the four sums are added up once per step --
  // both live in LDS (own lanes only, no synchronisation), the instance stays within the 128 VGPRs of 16 wavefronts per CU
  __shared__ T s_kt[RES ? 3 : 1][RES ? WT : 1][RES ? kPix : 1];
  __shared__ double s_acc[RES ? 4 : 1][RES ? WT : 1][RES ? kWave : 1];
  // column / row / plane indices: 32-bit in the plain instance -- the scalar unit orders 32-bit values itself, 64-bit orderings are
  // vector instructions (2048 x 2048 x 64, same box: 1.279 -> 1.233 ms per iteration).  The residual instance keeps 64-bit indices:
  // with 32-bit ones it spills 11 registers (none with these; 1.60 -> 2.06 ms); its counter form (RES with FLAGS) spills 12 either way
  // and runs 2.14 ms.
  typedef typename std::conditional<RES, long, int>::type idx_t;
  const idx_t nx = (idx_t)a.nx, ny = (idx_t)a.ny, L = (idx_t)a.L;
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));      // wave-uniform: plane, roles and base pointers live in SGPRs
  const unsigned groups = (unsigned)((L + P - 1) / P);
  const unsigned total = gridDim.x, chunks = a.chunks;
  const unsigned xcd = blockIdx.x % 8u, q = blockIdx.x / 8u;            // consecutive tiles on one XCD (its L2 serves what neighbours share)
  const unsigned tile = xcd * (total / 8u) + (xcd < total % 8u ? xcd : total % 8u) + q;
  // plane group fastest, then the row strip, then the column chunk: the workgroups that run side by side on an XCD share their
  // helper planes and the lines their halo lanes touch through its L2 (2048 x 2048 x 64: 6.2 GB read per launch; 6.6 GB with
  // the strip fastest, and 9 % more time with the strip slowest)
  const unsigned strips_n = total / (groups * chunks);
  const unsigned grp = tile % groups, strip = (tile / groups) % strips_n, chunk = tile / (strips_n * groups);
  const idx_t pl = (idx_t)grp * P - 1 + wv;              // this wavefront's plane (may lie outside the volume: idle, still takes part in the barriers)
  const bool exists = pl >= 0 && pl < L;
  const idx_t l = exists ? pl : 0;
  const bool do_y1 = exists && wv <= P + 1;
  const bool do_x2 = exists && wv >= 1 && wv <= P + 1;
  const bool do_y2 = exists && wv >= 1 && wv <= P;
  const bool has_above = exists && l + 1 < L, has_below = exists && l > 0;
  const idx_t row0 = (idx_t)strip * kRowsPerWave + ((idx_t)lane - kHalo) * VEC;
  const bool active = exists && row0 >= 0 && row0 < ny;
  const bool owner = active && lane >= kHalo && lane < kWave - kHalo;
  const idx_t xa = (idx_t)chunk * a.cols_per_block;
  const idx_t xb = xa + a.cols_per_block < nx ? xa + a.cols_per_block : nx;
  const size_t Pn = (size_t)nx * (size_t)ny, N = Pn * (size_t)L, plane = (size_t)l * Pn;
  const bool tiny_is_zero = a.f_val[1] >= (T)kTinyIsZeroRadius;     // device_math.hpp: norm2_leq0_fast
  const T* y1p = y + plane; const T* y2p = y + N + plane; const T* y3p = y + 2 * N + plane;
  const T* xp = x + plane;
  const T* bp = GB ? a.g_ptr[1] + plane : nullptr;

  typedef ColX2<T, VEC, GB> Col;
  const unsigned voff = (unsigned)(row0 * (idx_t)sizeof(T));          // meaningful in active lanes only (the others do not store)
  // Loads are UNCONDITIONAL: a lane outside the image reads the strip's first rows, a column outside the chunk's range the nearest
  // valid one, a missing neighbour plane the own plane -- values nobody uses (every use is guarded by the predicate that would have
  // guarded the load), but a register set that is overwritten as a whole each step is dead before it, which is what lets the three
  // sets rotate without copies or re-zeroing
  // (the residual instance, which sits at the 128-register limit, keeps predicated loads into a zeroed set: kUncond = false)
  constexpr bool kUncond = !RES;
  const unsigned voff_ld = kUncond && !active ? 0u : voff;
  const T* zxp = kUncond && !has_above ? xp : xp + Pn;
  const T* y3mp = kUncond && !has_below ? y3p : y3p - Pn;
  auto has_col = [&](idx_t k) { return k >= 0 && k < nx && k <= xb + 1; };
  auto load_col = [&](idx_t c, Col& in) {
    if (!kUncond) {
      in = Col{};
      if (!(active && has_col(c))) return;
    }
    const idx_t cc = !kUncond ? c : (c < 0 ? 0 : (c >= nx ? nx - 1 : c));
    const size_t o = (size_t)cc * (size_t)ny;                         // wave-uniform
    ldx_o<T, VEC>(y1p + o, voff_ld, in.y1); ldx_o<T, VEC>(y2p + o, voff_ld, in.y2); ldx_o<T, VEC>(y3p + o, voff_ld, in.y3); ldx_o<T, VEC>(xp + o, voff_ld, in.x);
    if constexpr (GB) ldx_o<T, VEC>(bp + o, voff_ld, in.b);
    if (kUncond || has_above) ldx_o<T, VEC>(zxp + o, voff_ld, in.zx);
    if (kUncond || has_below) ldx_o<T, VEC>(y3mp + o, voff_ld, in.y3m);
  };
  // primal step at column c of this plane (backend_pdhg.cu:317-338 with block_gradient3d.cu:127-149 on a zero-filled result);
  // v1 / v2 / v3: the dual variable at column c, p1c: its first component at column c-1, v3m: its third component one plane below
  auto primal = [&](idx_t c, const T (&v1)[VEC], const T (&v2)[VEC], const T (&v3)[VEC], const T (&p1c)[VEC], const T (&v3m)[VEC],
                    const T (&xin)[VEC], const T (&bv)[GB ? VEC : 1], const IterParams3<T>& Pm, T (&xn)[VEC], T (&ktv)[VEC]) {
    const T tauT = Pm.tau * a.Tval;
    const T up = lane_up(v2[VEC - 1]);                 // lane 0: no source, its first row is halo
    if constexpr (FMAD) {
      const T rD = (T)Pm.sq.rD;
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const idx_t row = row0 + j;
        const T divy = ((row < ny - 1) ? v2[j] : (T)0) - ((row > 0) ? (j > 0 ? v2[j > 0 ? j - 1 : 0] : up) : (T)0);
        const T divx = ((c < nx - 1) ? v1[j] : (T)0) - ((c > 0) ? p1c[j] : (T)0);
        const T divl = v3[j] - (has_below ? v3m[j] : (T)0);
        const T sdiv = divx + divy + divl;
        ktv[j] = -sdiv;
        const T arg = t_fma(tauT, sdiv, xin[j]);
        const T bj = GB ? bv[GB ? j : 0] : a.g_val[1];
        if (GFN == PROST_FN_SQUARE) xn[j] = t_fma(arg - bj, rD, bj);
        else { const T v = arg - bj; xn[j] = (v - t_max(t_min(v, Pm.step), -Pm.step)) + bj; }
      }
      return;
    }
    T parg[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const idx_t row = row0 + j;
      T divy = (row < ny - 1) ? v2[j] : (T)0;
      if (row > 0) divy -= (j > 0 ? v2[j > 0 ? j - 1 : 0] : up);
      T divx = (c < nx - 1) ? v1[j] : (T)0;
      if (c > 0) divx -= p1c[j];
      T divl = v3[j];
      if (has_below) divl -= v3m[j];
      const T kty = (T)0 - (divx + divy + divl);
      ktv[j] = kty;
      const T arg = xin[j] - tauT * kty;
      parg[j] = arg - (GB ? bv[GB ? j : 0] : a.g_val[1]);
    }
    // ElemOperation1D<F> with scalar a = 1, d = e = 0 (host-checked): the scaled prox is F_prox(v - b; step) + b.
    // F = Function1DSquare: the exact division; F = Function1DAbs (TV-L1 data term): soft threshold by step = c tau
    T r[VEC];
    if (GFN == PROST_FN_SQUARE) div_to_float_exact_vec<VEC>(parg, Pm.sq, r);
    else {
#pragma unroll
      for (int j = 0; j < VEC; j++) r[j] = f1d_apply<T, GFN>(a.g_fn, parg[j], Pm.step, a.g_val[5], a.g_val[6]);
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) xn[j] = r[j] + (GB ? bv[GB ? j : 0] : a.g_val[1]);
  };
  // K x (new iterate) and K x_prev (old iterate) of pixel j at column c (block_gradient3d.cu:62-80)
  auto gradients = [&](idx_t c, int j, const T (&xn_c)[VEC], const T (&xn_n)[VEC], const T (&xn_z)[VEC], const T (&xo_c)[VEC], const T (&xo_n)[VEC],
                       const T (&xo_z)[VEC], T bel_n, T bel_o, T (&kx)[3], T (&kp)[3]) {
    const idx_t row = row0 + j;
    const bool has_next = c + 1 < nx;
    const T below_n = (j < VEC - 1) ? xn_c[j < VEC - 1 ? j + 1 : 0] : bel_n;
    const T below_o = (j < VEC - 1) ? xo_c[j < VEC - 1 ? j + 1 : 0] : bel_o;
    kx[0] = has_next ? xn_n[j] - xn_c[j] : (T)0;
    kx[1] = (row < ny - 1) ? below_n - xn_c[j] : (T)0;
    kx[2] = has_above ? xn_z[j] - xn_c[j] : -xn_c[j];                          // Dirichlet (:73-76)
    kp[0] = has_next ? xo_n[j] - xo_c[j] : (T)0;
    kp[1] = (row < ny - 1) ? below_o - xo_c[j] : (T)0;
    kp[2] = has_above ? xo_z[j] - xo_c[j] : -xo_c[j];
  };
  // dual step at column c of this plane (backend_pdhg.cu:341-370): xn_* the new primal iterate at columns c / c+1 / one plane
  // above, xo_* the old one, v* the dual variable at column c
  auto dual = [&](idx_t c, const T (&xn_c)[VEC], const T (&xn_n)[VEC], const T (&xn_z)[VEC], const T (&xo_c)[VEC], const T (&xo_n)[VEC],
                  const T (&xo_z)[VEC], const T (&v1)[VEC], const T (&v2)[VEC], const T (&v3)[VEC], const IterParams3<T>& Pm,
                  T (&out)[3][VEC]) {
    const T sigS = Pm.sigma * a.Sval, theta = Pm.theta;
    const T bel_n = lane_down(xn_c[0]);                // lane 63: no source, its last row is halo
    const T bel_o = lane_down(xo_c[0]);
    T av[3][VEC], nv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      T kx[3], kp[3];
      gradients(c, j, xn_c, xn_n, xn_z, xo_c, xo_n, xo_z, bel_n, bel_o, kx, kp);
      const T yv[3] = {v1[j], v2[j], v3[j]};
      if constexpr (FMAD) {
        const T opt = 1 + theta;
        T nsq = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const T arg = t_fma(sigS, t_fma(opt, kx[i], -(theta * kp[i])), yv[i]);
          nsq = i == 0 ? arg * arg : t_fma(arg, arg, nsq);
          av[i][j] = arg;
        }
        const T sc = t_min(a.f_val[1] * t_rsq(nsq), (T)1);        // radius b > 0 (host-checked); ||v|| = 0: b * inf -> factor 1 on a zero vector
#pragma unroll
        for (int i = 0; i < 3; i++) out[i][j] = av[i][j] * sc;
      } else {
      T norm = 0;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const T arg = yv[i] + sigS * ((1 + theta) * kx[i] - theta * kp[i]);     // backend_pdhg.cu:54-70
        norm += arg * arg;
        av[i][j] = arg;
      }
      nv[j] = norm;
      }
    }
    if constexpr (!FMAD) norm2_leq0_fast<T, 3, VEC>(nv, av, a.f_val[1], tiny_is_zero, out);
  };
  // residual sums of the second iteration (RES): the terms of fused_iter3d_kernel
  const T sqT = t_sqrt(a.Tval), sqS = t_sqrt(a.Sval);
  const SharedDivisor<T> div_tauT(p2.tau * sqT), div_sigS(p2.sigma * sqS);   // wave-uniform: exact quotients through one double reciprocal each
  const T inv_tauT = (T)1 / (p2.tau * sqT), inv_sigS = (T)1 / (p2.sigma * sqS);  // (FMAD: the tolerance-compared sums use these)
  if (RES) {
#pragma unroll
    for (int k = 0; k < 4; k++) s_acc[RES ? k : 0][RES ? wv : 0][RES ? lane : 0] = 0;   // primal diff^2, primal var^2, dual diff^2, dual var^2
  }
  auto accumulate = [&](int k, double v) { s_acc[RES ? k : 0][RES ? wv : 0][RES ? lane : 0] += v; };
  // dual_residual_transform (backend_pdhg.cu:73-94) at a column of stage C: xo / xn = x^(k+1) / x^(k+2), kt_prev = K^T y^k, kt = K^T y^(k+1)
  auto dual_residual = [&](const T (&xo)[VEC], const T (&xn)[VEC], const T (&kt_prev)[VEC], const T (&kt)[VEC]) {
    double dd = 0, dv = 0;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const T w_hat = FMAD ? t_fma(-sqT, kt_prev[j], (xo[j] - xn[j]) * inv_tauT) : div_tauT.div(xo[j] - xn[j]) - sqT * kt_prev[j];
      const T diff = FMAD ? t_fma(sqT, kt[j], w_hat) : w_hat + sqT * kt[j];
      dd += (double)(diff * diff); dv += (double)(w_hat * w_hat);
    }
    if (owner) { accumulate(2, dd); accumulate(3, dv); }
  };
  // primal_residual_transform (backend_pdhg.cu:97-120) at the column of stage D: v* = y^(k+1), out = y^(k+2)
  auto primal_residual = [&](idx_t c, const T (&xn_c)[VEC], const T (&xn_n)[VEC], const T (&xn_z)[VEC], const T (&xo_c)[VEC], const T (&xo_n)[VEC],
                             const T (&xo_z)[VEC], const T (&v1)[VEC], const T (&v2)[VEC], const T (&v3)[VEC], const T (&out)[3][VEC]) {
    const T theta = p2.theta;
    const T bel_n = lane_down(xn_c[0]);
    const T bel_o = lane_down(xo_c[0]);
    double pd = 0, pv = 0;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      T kx[3], kp[3];
      gradients(c, j, xn_c, xn_n, xn_z, xo_c, xo_n, xo_z, bel_n, bel_o, kx, kp);
      const T yv[3] = {v1[j], v2[j], v3[j]};
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const T z_hat = FMAD ? t_fma(sqS, t_fma(1 + theta, kx[i], -(theta * kp[i])), (yv[i] - out[i][j]) * inv_sigS)
                             : div_sigS.div(yv[i] - out[i][j]) + sqS * ((1 + theta) * kx[i] - theta * kp[i]);
        const T diff = FMAD ? t_fma(-sqS, kx[i], z_hat) : z_hat - sqS * kx[i];
        pd += (double)(diff * diff); pv += (double)(z_hat * z_hat);
      }
    }
    if (owner) { accumulate(0, pd); accumulate(1, pv); }
  };
  auto publish = [&](T (&buf)[WT][kPix], const T (&v)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; j++) buf[wv][j * kWave + lane] = v[j];             // [j][lane]: conflict-free banks
  };
  auto fetch = [&](const T (&buf)[WT][kPix], int w, T (&v)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; j++) v[j] = buf[w][j * kWave + lane];
  };

  Col in1 = {}, in2 = {}, pre = {};                    // raw columns c+1, c+2 and the one being prefetched (c+3)
  // (tried in round 6 with the tolerance-class instance, which needs 10-15 registers less: a SECOND column in flight (c+4).  As a third
  // set copied down the chain nothing changes -- the copy itself waits for the newest loads (2.62 ms per launch either way at 2048 x
  // 2048 x 64); as two sets that take turns in a twice-unrolled loop, 20-27 registers spill and the launch takes 4.64 ms.)
  T b_c[GB ? VEC : 1];                                 // b of prox_g at column c (stage C)
  T x1_m[VEC], x1_0[VEC], x1_1[VEC], x1_2[VEC];        // x^(k+1) at columns c-1 .. c+2
  T xz1_m[VEC], xz1_0[VEC], xz1_1[VEC];                // x^(k+1) one plane above at columns c-1 .. c+1 (from LDS)
  T ya_m[VEC], yb_m[VEC], yc_m[VEC], ya_0[VEC], yb_0[VEC], yc_0[VEC], ya_1[VEC], yb_1[VEC], yc_1[VEC];   // y^(k+1) at columns c-1, c, c+1
  T x2_m[VEC], x2_0[VEC];                              // x^(k+2) at columns c-1, c
  T xz2_m[VEC];                                        // x^(k+2) one plane above at column c-1 (from LDS)
  T ym3_0[VEC];                                        // third component of y^(k+1) one plane below at column c (from LDS)
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    x1_m[j] = x1_0[j] = x1_1[j] = x1_2[j] = 0; xz1_m[j] = xz1_0[j] = xz1_1[j] = 0;
    ya_m[j] = yb_m[j] = yc_m[j] = ya_0[j] = yb_0[j] = yc_0[j] = ya_1[j] = yb_1[j] = yc_1[j] = 0;
    x2_m[j] = x2_0[j] = 0; xz2_m[j] = 0; ym3_0[j] = 0;
  }
#pragma unroll
  for (int j = 0; j < (GB ? VEC : 1); j++) b_c[j] = 0;
  if (active) {
    if (xa - 2 >= 0) ldx_o<T, VEC>(y1p + (size_t)(xa - 2) * (size_t)ny, voff, in2.y1);   // becomes in1.y1 for A(xa-1)
  }
  if (exists || !kUncond) load_col(xa - 1, pre);

  // nothing of the prologue may still be in flight when the loop starts: the compiler's wait-count insertion otherwise carries
  // the prologue's pending loads around the loop and drains vmcnt to 0 INSIDE every step -- which also waits for the column that
  // was just prefetched
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
  if (FLAGS) {
    if (lane == 0) s_step[FLAGS ? wv : 0] = -1;
    __syncthreads();
  }
  int k3 = 0;                                          // slot of s_kt that stage A writes in this step
  int step = 0, wr3 = 0;                               // FLAGS: step counter and its slot (step % 3)
  // (tried: the loop unrolled three times with the three raw-column sets renamed instead of copied -- 28 moves less per step,
  // 40-80 spilled registers at the 128 the instance may use)
  // (tried in round 4: a second instance of the column step for the steady state -- columns c-1 .. c+3 inside the chunk's range and
  // strictly inside the image, so no column select is left in the stencils and every stage condition is the wavefront's role alone:
  // 16 selects and a dozen scalar compares less per step, but the two copies together no longer fit 128 VGPRs (15 spilled))
  // one column step; `cur` holds raw column c+2 and is refilled with the column that will be needed when its turn comes again
  auto column_step = [&](idx_t c, Col& cur) {
    const int wr = FLAGS ? wr3 : (int)((c + 4) & 1), rd = FLAGS ? (wr3 == 0 ? 2 : wr3 - 1) : wr ^ 1;
    if (FLAGS && step > 0) {
      // wait for the neighbouring planes' wavefronts (whether or not their planes exist: every wavefront counts its steps)
      if (wv >= 1) while (__builtin_amdgcn_readfirstlane(lds_flag_read(&s_step[FLAGS ? wv - 1 : 0])) < step - 1) __builtin_amdgcn_s_sleep(1);
      if (wv + 1 < WT) while (__builtin_amdgcn_readfirstlane(lds_flag_read(&s_step[FLAGS ? wv + 1 : 0])) < step - 1) __builtin_amdgcn_s_sleep(1);
    }
    // raw columns: in1 <- in2 <- cur, prefetch column c+3
    in1 = in2; in2 = cur;
    if (exists || !kUncond) load_col(c + 3, cur);
    // neighbour-plane values published in the previous step
    if (has_above && wv + 1 < WT) { fetch(s_x1[rd], wv + 1, xz1_1); fetch(s_x2[rd], wv + 1, xz2_m); }
    if (has_below && wv >= 1) fetch(s_y3[rd], wv - 1, ym3_0);
    const idx_t ca = c + 2, cb = c + 1, cd = c - 1;
    if (exists && ca >= (xa - 1 > 0 ? xa - 1 : 0) && ca < nx && ca <= xb + 1) {                    // stage A
      T kt_a[VEC];
      primal(ca, in2.y1, in2.y2, in2.y3, in1.y1, in2.y3m, in2.x, in2.b, p1, x1_2, kt_a);
      if (RES) {
#pragma unroll
        for (int j = 0; j < VEC; j++) s_kt[RES ? k3 : 0][RES ? wv : 0][RES ? j * kWave + lane : 0] = kt_a[j];       // K^T y^k at column c+2: read by stage C two steps later
      }
      publish(s_x1[wr], x1_2);
    }
    if (do_y1 && cb >= (xa - 1 > 0 ? xa - 1 : 0) && cb < nx && cb <= xb) {                         // stage B
      T o[3][VEC];
      dual(cb, x1_1, x1_2, xz1_1, in1.x, in2.x, in1.zx, in1.y1, in1.y2, in1.y3, p1, o);
#pragma unroll
      for (int j = 0; j < VEC; j++) { ya_1[j] = o[0][j]; yb_1[j] = o[1][j]; yc_1[j] = o[2][j]; }
      publish(s_y3[wr], yc_1);
    }
    if (do_x2 && c >= xa && c < nx && c <= xb) {                                                  // stage C
      T kt_c[VEC];
      primal(c, ya_0, yb_0, yc_0, ya_m, ym3_0, x1_0, b_c, p2, x2_0, kt_c);
      publish(s_x2[wr], x2_0);
      if (RES && do_y2 && c < xb) {
        T kt_0[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) kt_0[j] = s_kt[RES ? (k3 + 1) % 3 : 0][RES ? wv : 0][RES ? j * kWave + lane : 0];     // written two steps ago
        dual_residual(x1_0, x2_0, kt_0, kt_c);
      }
      if (do_y2 && owner && c < xb) stx_o<T, VEC>(x_out + plane + (size_t)c * (size_t)ny, voff, x2_0);
    }
    if (do_y2 && cd >= xa && cd < xb) {                                                           // stage D
      T o[3][VEC];
      dual(cd, x2_m, x2_0, xz2_m, x1_m, x1_0, xz1_m, ya_m, yb_m, yc_m, p2, o);
      if (RES) primal_residual(cd, x2_m, x2_0, xz2_m, x1_m, x1_0, xz1_m, ya_m, yb_m, yc_m, o);
      if (owner) {
        const size_t off = plane + (size_t)cd * (size_t)ny;           // wave-uniform
        stx_o<T, VEC>(y_out + off, voff, o[0]); stx_o<T, VEC>(y_out + N + off, voff, o[1]); stx_o<T, VEC>(y_out + 2 * N + off, voff, o[2]);
      }
    }
    if (FLAGS) {
      if (lane == 0) lds_flag_write(&s_step[FLAGS ? wv : 0], step);          // after this step's publishes (LDS is in order per wavefront)
      step++; wr3 = wr3 == 2 ? 0 : wr3 + 1;
    } else {
      __syncthreads();                                 // one barrier per column: the buffers written now are read in the next step
    }
    // shift the pipeline by one column
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      x1_m[j] = x1_0[j]; x1_0[j] = x1_1[j]; x1_1[j] = x1_2[j];
      xz1_m[j] = xz1_0[j]; xz1_0[j] = xz1_1[j];
      ya_m[j] = ya_0[j]; yb_m[j] = yb_0[j]; yc_m[j] = yc_0[j];
      ya_0[j] = ya_1[j]; yb_0[j] = yb_1[j]; yc_0[j] = yc_1[j];
      x2_m[j] = x2_0[j];
    }
#pragma unroll
    for (int j = 0; j < (GB ? VEC : 1); j++) b_c[j] = in1.b[j];
    k3 = k3 == 2 ? 0 : k3 + 1;
  };
  for (idx_t c = xa - 3; c <= xb; c++) column_step(c, pre);
  if (RES) {                                           // the last barrier of the loop has passed: the exchange buffers are free
    if (FLAGS) __syncthreads();
    double* sred = reinterpret_cast<double*>(&s_x1[0][0][0]);
    double r_pd = s_acc[0][RES ? wv : 0][RES ? lane : 0], r_pv = s_acc[RES ? 1 : 0][RES ? wv : 0][RES ? lane : 0], r_dd = s_acc[RES ? 2 : 0][RES ? wv : 0][RES ? lane : 0],
           r_dv = s_acc[RES ? 3 : 0][RES ? wv : 0][RES ? lane : 0];
    r_pd = wave_sum(r_pd); r_pv = wave_sum(r_pv); r_dd = wave_sum(r_dd); r_dv = wave_sum(r_dv);
    if (lane == 0) { sred[4 * wv + 0] = r_pd; sred[4 * wv + 1] = r_pv; sred[4 * wv + 2] = r_dd; sred[4 * wv + 3] = r_dv; }
    __syncthreads();
    if (threadIdx.x < 4) {
      double t = 0;
      for (int w = 0; w < WT; w++) t += sred[4 * w + threadIdx.x];       // fixed order: run-to-run deterministic
      partial[4 * (size_t)blockIdx.x + threadIdx.x] = t;
    }
  }
}

bool fused3d_desc_ok(const prost_hip_fused_desc* d);

// rows per lane / wavefronts per workgroup of the instance that runs (see the header of this file)
// (8 bytes of rows per lane: 2 floats or 1 double)
constexpr int kX2Waves = 16;
// rows per lane: 2 floats where the height is even, otherwise (and for doubles) 1
static int x2_vec(int dtype, size_t ny) { return dtype == 0 && ny % 2 == 0 ? 2 : 1; }
static size_t x2_rows_per_wave(int vec) { return vec == 2 ? (size_t)(kWave - 2) * 2 : (size_t)(kWave - 4); }      // owned rows: one halo lane of 2 rows / two of 1 row on either side

// straight-line ROF / TV-L1 shapes
static bool iter3d_x2_ok(const prost_hip_fused_desc* d, int dtype) {
  if ((dtype != 0 && dtype != 1) || !fused3d_desc_ok(d)) return false;
  if (d->ny < 4 || d->nx < 4) return false;
  if ((d->g_fn != PROST_FN_SQUARE && d->g_fn != PROST_FN_ABS) || d->f_fn != PROST_FN_IND_LEQ0) return false;
  for (int k = 0; k < 7; k++) {
    if (d->f_coeff_ptr[k]) return false;
    if (k != 1 && d->g_coeff_ptr[k]) return false;
  }
  if (d->g_coeff_ptr[1] && !aligned16(d->g_coeff_ptr[1])) return false;
  if (d->g_coeff_val[0] != 1.0 || d->g_coeff_val[2] == 0.0 || d->g_coeff_val[3] != 0.0 || d->g_coeff_val[4] != 0.0) return false;
  if (d->f_coeff_val[0] != 1.0 || d->f_coeff_val[3] != 0.0 || d->f_coeff_val[4] != 0.0) return false;
  if (d->res_x1 != 0 && !(d->res_x0 == 0 && d->res_x1 >= d->nx)) return false;
  const size_t rows = x2_rows_per_wave(x2_vec(dtype, d->ny));
  const size_t strips = (d->ny + rows - 1) / rows;
  // (residual launches write one partial per workgroup: with one chunk per tile the tiles alone must fit the reduction workspace)
  const size_t tiles = strips * ((d->L + kX2Waves - 4) / (kX2Waves - 3));
  return tiles <= (size_t)kReduceBlocks / 2 && tiles * d->nx < (size_t)1 << 31;
}

// tolerance-class instances: fp32, 2 rows per lane (even heights), ind_leq0 radius > 0; anything else computes exactly
static bool iter3d_x2_fmad(const prost_hip_fused_desc* d, int dtype) {
  return d->arith == PROST_HIP_ARITH_FMAD && dtype == 0 && iter3d_x2_ok(d, dtype) && x2_vec(dtype, d->ny) == 2 && d->f_coeff_val[1] > 0.0;
}

static int compute_units() {
  static int n = 0;
  if (n == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount; else n = 256;
  }
  return n;
}

// One workgroup per compute unit is resident (16 wavefronts x 121-128 VGPRs) and all workgroups take equally long, so a launch lasts
// (rounds of workgroups) x (column steps of a workgroup): pick the number of column chunks that minimises
// ceil(strips * groups * chunks / CUs) * (columns per chunk + 4 warm-up steps).  2048 x 2048 x 64 on 256 CUs: 17 x 5 x 3 = 255
// workgroups, ONE round of 687 steps (64-column chunks: 11 rounds of 68 steps = 748).
static size_t x2_chunk_cols(const prost_hip_fused_desc* d, int dtype, int cols, bool res) {
  constexpr int P = kX2Waves - 3;
  const size_t rows = x2_rows_per_wave(x2_vec(dtype, d->ny));
  const size_t tiles = ((d->ny + rows - 1) / rows) * ((d->L + P - 1) / P);
  const size_t max_groups = res ? (size_t)kReduceBlocks / 2 : (size_t)1 << 31;     // residual launches: one partial (4 doubles) per workgroup
  if (cols > 0) return (size_t)cols < d->nx ? (size_t)cols : d->nx;
  const size_t cus = (size_t)compute_units();
  size_t best_c = d->nx, best_cost = (size_t)-1;
  for (size_t n = 1; n <= d->nx; n++) {
    const size_t c = (d->nx + n - 1) / n;
    if (c < 8 && n > 1) break;
    if (tiles * ((d->nx + c - 1) / c) > max_groups && n > 1) break;
    const size_t cost = ((tiles * ((d->nx + c - 1) / c) + cus - 1) / cus) * (c + 4);
    if (cost < best_cost) { best_cost = cost; best_c = c; }
  }
  return best_c;
}

template <class T, int V, int WT>
static int launch_iter3d_x2(const prost_hip_fused_desc* d, T* x_out, T* y_out, const T* x, const T* y, const IterParams3<T> (&p)[2], int cols,
                            double* out4, void* ws, hipStream_t s, void* record = nullptr, const RuleTail* tail = nullptr) {
  constexpr int P = WT - 3;
  FusedArgs<T> a = make_fused_args<T>(d);
  const size_t groups = (d->L + P - 1) / P;
  const size_t strips = (d->ny + x2_rows_per_wave(V) - 1) / x2_rows_per_wave(V);
  const size_t c = x2_chunk_cols(d, std::is_same<T, float>::value ? 0 : 1, cols, out4 != nullptr);
  a.cols_per_block = (int)c;
  a.chunks = (unsigned)((d->nx + c - 1) / c);
  const unsigned grid = (unsigned)(strips * a.chunks * groups);
  if (out4 && grid > (unsigned)kReduceBlocks / 2) { set_error("fused 3-D double iteration: grid exceeds the reduction workspace"); return 1; }
  double* partial = static_cast<double*>(ws);
  static const bool flags = []() { const char* e = getenv("PROST_X2_SYNC"); return !(e && atoi(e) == 0); }();      // PROST_X2_SYNC=0: the barrier per step (A/B)
  const bool fmad = iter3d_x2_fmad(d, std::is_same<T, float>::value ? 0 : 1);
#define GO4(G, B, R, F, A) PH_LAUNCH((fused_iter3d_x2_kernel<T, V, G, B, WT, R, F, A>), dim3(grid), dim3(kWave * WT), 0, s, x_out, y_out, x, y, a, p[0], \
    p[1], partial, static_cast<const PdhgRecord<T>*>(record))
#define GO3(G, B, R, F) do { bool done_ = false; if constexpr (std::is_same<T, float>::value && V == 2) { if (fmad) { GO4(G, B, R, F, true); done_ = true; \
    } } if (!done_) GO4(G, B, R, F, false); } while (0)
#define GO2(G, B, R) do { if (flags && !R) GO3(G, B, false, true); else GO3(G, B, R, false); } while (0)
#define GO(B, R) do { if (d->g_fn == PROST_FN_ABS) GO2(PROST_FN_ABS, B, R); else GO2(PROST_FN_SQUARE, B, R); } while (0)
  if (d->g_coeff_ptr[1]) { if (out4) GO(true, true); else GO(true, false); }
  else { if (out4) GO(false, true); else GO(false, false); }
#undef GO
#undef GO2
#undef GO3
#undef GO4
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "fused 3-D double iteration kernel"); }
  if (out4 && tail && tail->apply) return launch_fold4_rule<T>(out4, partial, grid, record, tail->iteration, tail->mirror, s);
  if (out4) return launch_fold4(out4, partial, grid, s);
  return 0;
}

template <class T>
static int run_iter3d_x2(const prost_hip_fused_desc* d, T* x_out, T* y_out, const T* x, const T* y, const double* tau, const double* sigma,
                         const double* theta, int cols, double* out4, void* ws, void* stream, void* record = nullptr, const RuleTail* tail = nullptr) {
  constexpr int kDtype = std::is_same<T, float>::value ? 0 : 1;
  if (!iter3d_x2_ok(d, kDtype)) { set_error("fused 3-D double iteration: unsupported description (see prost_hip_fused_iteration3d_x2_supported)"); return 1; }
  if (!aligned16(x_out) || !aligned16(y_out) || !aligned16(x) || !aligned16(y)) { set_error("fused 3-D double iteration: vectors must be 16-byte aligned"); return 1; }
  if (x_out == x || y_out == y) { set_error("fused 3-D double iteration: outputs must not alias inputs"); return 1; }
  if (out4 && !ws) { set_error("fused 3-D double iteration: residuals need the reduction workspace"); return 1; }
  FusedArgs<T> a = make_fused_args<T>(d);
  IterParams3<T> p[2];
  for (int i = 0; i < 2; i++) {
    p[i].tau = (T)tau[i]; p[i].sigma = (T)sigma[i]; p[i].theta = (T)theta[i];
    const UniformProx<T> ug = make_uniform_prox<T>(a.g_val, (T)tau[i] * a.Tval);
    const UniformProx<T> uf = make_uniform_prox<T>(a.f_val, (T)sigma[i] * a.Sval);
    if (!ug.a_one || !ug.den_one || ug.degenerate || !uf.a_one || !uf.den_one) { set_error("fused 3-D double iteration: not the straight-line ROF shape"); return 1; }
    p[i].sq = ug.sq; p[i].step = ug.step;
  }
  if constexpr (std::is_same<T, float>::value) {
    if (x2_vec(kDtype, d->ny) == 2) return launch_iter3d_x2<T, 2, kX2Waves>(d, x_out, y_out, x, y, p, cols, out4, ws, as_stream(stream), record, tail);
  }
  return launch_iter3d_x2<T, 1, kX2Waves>(d, x_out, y_out, x, y, p, cols, out4, ws, as_stream(stream), record, tail);
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_fused_iteration3d_x2_supported(const prost_hip_fused_desc* d, int dtype) { return iter3d_x2_ok(d, dtype) ? 1 : 0; }
int prost_hip_fused_iteration3d_x2_arith(const prost_hip_fused_desc* d, int dtype) { return iter3d_x2_fmad(d, dtype) ? PROST_HIP_ARITH_FMAD : PROST_HIP_ARITH_EXACT; }
int prost_hip_fused_iteration3d_x2_chunk_cols(const prost_hip_fused_desc* d, int dtype, int with_residuals) {
  return iter3d_x2_ok(d, dtype) ? (int)x2_chunk_cols(d, dtype, 0, with_residuals != 0) : 0;
}
int prost_hip_fused_iteration3d_x2_f32(const prost_hip_fused_desc* d, float* x_out, float* y_out, const float* x, const float* y, const double* tau,
                                       const double* sigma, const double* theta, int cols, double* res_out4, void* workspace, void* stream) {
  return run_iter3d_x2<float>(d, x_out, y_out, x, y, tau, sigma, theta, cols, res_out4, workspace, stream);
}
int prost_hip_fused_iteration3d_x2_f64(const prost_hip_fused_desc* d, double* x_out, double* y_out, const double* x, const double* y, const double* tau,
                                       const double* sigma, const double* theta, int cols, double* res_out4, void* workspace, void* stream) {
  return run_iter3d_x2<double>(d, x_out, y_out, x, y, tau, sigma, theta, cols, res_out4, workspace, stream);
}
// both iterations with the step sizes of the device record (prost_hip_pdhg_rule_begin); apply_rule: the fold of the residual sums also
// evaluates the step-size rule and the stopping test
int prost_hip_fused_iteration3d_x2_rec_f32(const prost_hip_fused_desc* d, float* x_out, float* y_out, const float* x, const float* y, void* record, int cols,
                                           double* res_out4, void* workspace, int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!record) { set_error("fused_iteration3d_x2_rec: no record"); return 1; }
  const double one[2] = {1.0, 1.0};
  const RuleTail tail{apply_rule, iteration, mirror};
  return run_iter3d_x2<float>(d, x_out, y_out, x, y, one, one, one, cols, res_out4, workspace, stream, record, &tail);
}
int prost_hip_fused_iteration3d_x2_rec_f64(const prost_hip_fused_desc* d, double* x_out, double* y_out, const double* x, const double* y, void* record, int cols,
                                           double* res_out4, void* workspace, int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void* stream) {
  if (!record) { set_error("fused_iteration3d_x2_rec: no record"); return 1; }
  const double one[2] = {1.0, 1.0};
  const RuleTail tail{apply_rule, iteration, mirror};
  return run_iter3d_x2<double>(d, x_out, y_out, x, y, one, one, one, cols, res_out4, workspace, stream, record, &tail);
}
}  // extern "C"
