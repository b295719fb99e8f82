// kernels_fused_iterk.hip -- K PDHG iterations per kernel launch (K = 2, 3, 4), gradient2d, one channel, fp32,
// tolerance-class arithmetic (prost_hip_fused_desc.arith = PROST_HIP_ARITH_FMAD).
//
// The pair kernel (kernels_fused_iter2.hip) in its tolerance-class instance is no longer bound by instruction issue: it moves
// its ~565 MB per launch at the rate a copy kernel reaches.  The only way further down is fewer bytes per ITERATION: this
// kernel keeps x^(k+1) .. x^(k+K-1), y^(k+1) .. y^(k+K-1) on chip and reads x^k, y^k, b once and writes x^(k+K), y^(k+K) --
// 7 floats per pixel per K iterations (backend_pdhg.cu:313-381 K times; block_gradient2d.cu:61-77, :122-138 inlined).
//
// A wavefront marches over a chunk of columns with a 2K-stage software pipeline, all in registers.  In step c
//     P_l : x^l at column c + K - l + 1   (needs x^(l-1), y^(l-1) there and y_1^(l-1) one column to the left)
//     D_l : y^l at column c + K - l       (needs y^(l-1) there, x^l and x^(l-1) there and one column to the right)
// for l = 1 .. K in that order; level 0 is the loaded iterate.  Rows: a lane owns 4 consecutive rows, row neighbours come from the
// adjacent lanes (DPP shifts).  Every iteration invalidates one more row at the top and at the bottom of the wave's 256 rows:
// H = ceil(K / 4) HALO lanes per side are never stored, the wave owns (64 - 2H) x 4 rows (62 x 4 for K <= 4).  Columns: a chunk
// starts its pipeline 2K - 1 steps early and loads K - 1 columns (plus one column of y_1) before and K - 1 columns after its own.
//
// Loads run through an LDS ring of R slots fed by LDS-DMA (global_load_lds_dwordx4, no VGPRs), R - 1 columns ahead, and from the
// ring into a register stage one step ahead: the ds_reads of the NEXT step's column are issued in the middle of a step's arithmetic
// and waited for at the top of the next step, so neither the HBM nor the LDS latency is exposed.  vmcnt retires in issue order, loads
// and stores alike, so the wait in front of the ring read is COUNTED: in the steady state (the steps since the needed batch stored
// their three vectors each) everything issued after that batch may stay in flight.
//
// Arithmetic: the tolerance-class forms of the pair kernel (fused multiply-adds, fp32 reciprocal of 1 + step, v_rsq_f32).  A
// K-launch equals K/2 pair launches of that class bit for bit (same expressions, same order; tests/test_gpu_fmad.py).
#include "fused_common.hpp"
#include "reduce.hpp"

#include <type_traits>

#ifndef ITERK_NT
#define ITERK_NT true
#endif
#ifndef ITERK_LOAD_AUX
#define ITERK_LOAD_AUX 0
#endif

namespace prost_hip {

typedef int idx_t;
typedef __attribute__((address_space(3))) void lds_void_k;
typedef const __attribute__((address_space(1))) void glb_void_k;

struct LevelStep {
  float tauT, rD, step, sigS, theta, opt;       // tau T, 1 / (1 + step), step = c a^2 tau T, sigma S, theta, 1 + theta
  UniformDiv sq;                                // exact class: the divisor 1. + step of Function1DSquare and its reciprocal, in double
};
template <int K>
struct StepsK {
  LevelStep s[K];
  float tau_last, sigma_last;                   // residual transforms of the last iteration
};

template <int N> using int_c = std::integral_constant<int, N>;
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (A < B) { f(int_c<A>()); static_for<A + 1, B>(f); }
}

template <int R, int NB, int STORES>
__device__ __forceinline__ void ring_wait(bool steady, bool full) {
  // issue order of a step: [x store] [ring wait + read of the next column] [LDS-DMA batch] [y stores].  Behind the batch about to be
  // read lie the y stores of the step that issued it, then per step an x store, a batch and the y stores, then this step's x store.
  // `full`: R - 2 batches were issued after the one about to be read; `steady`: and every step since stored its STORES vectors
  constexpr int kCons = (R - 2) * NB, kSteady = (R - 2) * NB + (R - 1) * STORES;
  static_assert(kSteady <= 63, "vmcnt is a 6-bit counter");
  if (steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kSteady) : "memory");
  else if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kCons) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// GFN: Function1DSquare or Function1DAbs (scalar a = 1, d = e = 0, c != 0), BPTR: b of prox_g per pixel, RES: the four residual
// sums of the LAST iteration, R: ring slots, WAVES: resident wavefronts per SIMD the register budget is cut for
// EXACT: the arithmetic of the exact class -- the expressions of kernels_fused_iter2.hip's straight-line instances, operation for
// operation (no contraction, the correctly rounded quotient and root forms of device_math.hpp): a launch then equals the CPU oracle bit for
// bit, like the pair kernel -- which pins the pipeline itself (stage order, halo lanes, ring, chunk edges) to the oracle, independently of
// the tolerance-class arithmetic.  The exact forms are bound by instruction issue (~71 VALU per pixel-iteration against ~17): measured at
// 4096^2, K = 2 / 3 / 4: 0.106 / 0.145 / 0.204 ms per launch = 53 / 48 / 51 us per iteration, no better than the pair kernel, so the host
// keeps pairs for the exact class (BackendPDHG::Options::group_max asks for groups explicitly).
template <int K, int GFN, bool BPTR, bool RES, int R, int WAVES, bool EXACT>
__global__ void __launch_bounds__(kWave, WAVES)
    fused_iter2d_xk_kernel(float* __restrict__ x_out, float* __restrict__ y_out, const float* __restrict__ x, const float* __restrict__ y,
                           FusedArgs<float> a, StepsK<K> sp, double* __restrict__ partial, const PdhgRecord<float>* __restrict__ rec) {
  static_assert(K >= 1 && K <= 6 && R >= 2, "register budget: K <= 6");
  constexpr int VEC = 4;
  constexpr int H = (K + VEC - 1) / VEC;          // halo lanes per side
  typedef float T;
  if (rec) {                       // device-resident step sizes: the same for every iteration of the launch (no rule evaluation inside)
    if (rec->stop) return;
    LevelStep s;
    s.tauT = rec->p.tau * a.Tval; s.rD = (float)rec->p.ug.sq.rD; s.step = rec->p.ug.step; s.sigS = rec->p.sigma * a.Sval;
    s.theta = rec->p.theta; s.opt = 1 + rec->p.theta; s.sq = rec->p.ug.sq;
#pragma unroll
    for (int l = 0; l < K; l++) sp.s[l] = s;
    sp.tau_last = rec->p.tau; sp.sigma_last = rec->p.sigma;
  }
  const idx_t nx = (idx_t)a.nx, ny = (idx_t)a.ny;
  const idx_t rx0 = (idx_t)a.rx0, rx1 = a.rx1 > (size_t)0x7fffffff ? (idx_t)0x7fffffff : (idx_t)a.rx1;
  const int lane = threadIdx.x;
  constexpr int kRowsPerWave = (kWave - 2 * H) * VEC;
  const unsigned total = gridDim.x, chunks = a.chunks;
  const unsigned xcd = blockIdx.x % 8u, q8 = blockIdx.x / 8u;          // XCD-aware tile order (kernels_fused_iter.hip)
  const unsigned tile = xcd * (total / 8u) + (xcd < total % 8u ? xcd : total % 8u) + q8;
  // a.strips != 0: the strips of one chunk are consecutive tiles -- vertical neighbours run side by side on one XCD, march over the same
  // columns at the same time and meet in its L2 on the lines their halo rows share (a strip is 7.75 lines high)
  const unsigned strip = a.strips ? tile % a.strips : tile / chunks, chunk = a.strips ? tile / a.strips : tile % chunks;
  const idx_t row0 = (idx_t)strip * kRowsPerWave + ((idx_t)lane - H) * VEC;
  const bool active = row0 >= 0 && row0 < ny;
  const bool owner = active && lane >= H && lane < kWave - H;
  const idx_t xa = (idx_t)chunk * a.cols_per_block;
  const idx_t xb = xa + a.cols_per_block < nx ? xa + a.cols_per_block : nx;
  const size_t N = (size_t)nx * (size_t)ny;
  constexpr int NB = 3 + (BPTR ? 1 : 0);
  __shared__ __attribute__((aligned(16))) char ring_mem[R * NB * 1024];
  const T* const y2base = y + N;
  T* const y2out = y_out + N;
  const T* const bptr = BPTR ? a.g_ptr[1] : nullptr;
  const T bval = a.g_val[1], bq = a.f_val[1];
  const bool tiny_is_zero = a.f_val[1] >= (T)kTinyIsZeroRadius;     // device_math.hpp: norm2_leq0_fast
  auto off_of = [&](idx_t c) { return ((unsigned)c * (unsigned)ny + (unsigned)row0) * (unsigned)sizeof(T); };
  // loaded columns: xa - K + 1 .. xb + K - 1 (the y_1 of column xa - K is read directly into registers)
  auto has_col = [&](idx_t k) { return k >= 0 && k < nx && k >= xa - K + 1 && k <= xb + K - 1; };
  auto slot_of = [&](idx_t k) { return (int)((unsigned)(k + 4 * R) % (unsigned)R); };      // k >= -K > -4R
  auto ring_issue = [&](idx_t k) {
    if (active) {
      const unsigned o = off_of(k);
      char* slot = ring_mem + slot_of(k) * (NB * 1024);
      __builtin_amdgcn_global_load_lds((glb_void_k*)(reinterpret_cast<const char*>(y) + o), (lds_void_k*)(slot), 16, 0, ITERK_LOAD_AUX);
      __builtin_amdgcn_global_load_lds((glb_void_k*)(reinterpret_cast<const char*>(y2base) + o), (lds_void_k*)(slot + 1024), 16, 0, ITERK_LOAD_AUX);
      __builtin_amdgcn_global_load_lds((glb_void_k*)(reinterpret_cast<const char*>(x) + o), (lds_void_k*)(slot + 2048), 16, 0, ITERK_LOAD_AUX);
      if (BPTR) __builtin_amdgcn_global_load_lds((glb_void_k*)(reinterpret_cast<const char*>(bptr) + o), (lds_void_k*)(slot + 3072), 16, 0, ITERK_LOAD_AUX);
    }
  };
  typedef native_f4 V4;
  V4 n0 = {}, n1 = {}, n2 = {}, n3 = {};           // register stage: the column of the NEXT step (y_1, y_2, x, b), read from the ring one step ahead
  auto ring_read_async = [&](idx_t k) {            // four ds_read_b128, NOT waited for (ring_read_done)
    const unsigned addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)ring_mem) + (unsigned)slot_of(k) * (unsigned)(NB * 1024) + (unsigned)lane * 16u;
    asm volatile("ds_read_b128 %0, %1" : "=v"(n0) : "v"(addr) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(n1) : "v"(addr) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(n2) : "v"(addr) : "memory");
    if (BPTR) asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(n3) : "v"(addr) : "memory");
  };
  auto ring_read_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3)::"memory"); };
  // the register stage <- column k1; then the LDS-DMA batch of column k1 + R - 1 into the slot of column k1 - 1, whose read has been
  // waited for.  c = the step this runs in (k1 = c + K + 1): the steps c - R + 2 .. c - 1 and this one have stored before it.
  auto prefetch = [&](idx_t k1, idx_t c) {
    if (has_col(k1)) {
      bool full = true;
#pragma unroll
      for (int r = 1; r <= R - 2; r++) full = full && has_col(k1 + r);
      ring_wait<R, NB, 3>(full && c - R + 1 >= xa && c + 1 < xb, full);
      ring_read_async(k1);
    } else { n0 = V4{}; n1 = V4{}; n2 = V4{}; n3 = V4{}; }
    if (has_col(k1 + R - 1)) ring_issue(k1 + R - 1);
  };

  // ---- pipeline state -------------------------------------------------------------------------------------------------
  // X[l]: x^l at three consecutive columns, newest last (level 0: the two loaded columns c + K - 1, c + K in slots 1, 2;
  // level l >= 1: columns c + K - l - 1 .. c + K - l + 1).  Y[l]: y^l (both components) at two columns, newest last
  // (level 0: c + K - 1, c + K; level l >= 1: c + K - l - 1, c + K - l).  B[i]: b at column c + K - i.
  T X[K + 1][3][VEC], Y[K][2][2][VEC], B[K][VEC];
  T KTp[RES ? VEC : 1], KTc[RES ? VEC : 1], KTn[RES ? VEC : 1];   // K^T y^(K-2) at column c + 1 / K^T y^(K-1) at c + 1 / K^T y^(K-2) at c + 2
#pragma unroll
  for (int l = 0; l <= K; l++)
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
      for (int j = 0; j < VEC; j++) X[l][s][j] = 0;
#pragma unroll
  for (int l = 0; l < K; l++)
#pragma unroll
    for (int j = 0; j < VEC; j++) { Y[l][0][0][j] = Y[l][0][1][j] = Y[l][1][0][j] = Y[l][1][1][j] = 0; B[l][j] = 0; }
#pragma unroll
  for (int j = 0; j < (RES ? VEC : 1); j++) KTp[j] = KTc[j] = KTn[j] = 0;
  double r_pd = 0, r_pv = 0, r_dd = 0, r_dv = 0;
  const T sqT = t_sqrt(a.Tval), sqS = t_sqrt(a.Sval);
  const T inv_sigS = (T)1 / (sp.sigma_last * sqS), inv_tauT = (T)1 / (sp.tau_last * sqT);

  // primal step of level l at column c: x^l = prox_g(x^(l-1) - tau T K^T y^(l-1))
  auto primal = [&](auto inner, idx_t c, const LevelStep& P, const T (&y1c)[VEC], const T (&y2c)[VEC], const T (&y1p)[VEC], const T (&xin)[VEC],
                    const T (&bc)[VEC], T (&xn)[VEC], T (&kt)[RES ? VEC : 1], bool want_kt) {
    constexpr bool I = decltype(inner)::value;
    const T up = lane_up(y2c[VEC - 1]);            // lane 0: no source, its first row is halo
    T parg[EXACT ? VEC : 1];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const idx_t row = row0 + j;
      const T upj = (j > 0) ? y2c[(j + VEC - 1) % VEC] : up;
      const T divy = ((I || row < ny - 1) ? y2c[j] : (T)0) - ((I || row > 0) ? upj : (T)0);
      const T divx = ((I || c < nx - 1) ? y1c[j] : (T)0) - ((I || c > 0) ? y1p[j] : (T)0);
      const T sdiv = divx + divy;
      if constexpr (EXACT) {        // backend_pdhg.cu:317-338 as kernels_fused_iter2.hip evaluates it: K^T y = 0 - (div), x - tau T K^T y, v - b
        const T kty = (T)0 - sdiv;
        if (RES && want_kt) kt[RES ? j : 0] = kty;
        const T arg = xin[j] - P.tauT * kty;
        parg[EXACT ? j : 0] = arg - bc[j];
      } else {
        if (RES && want_kt) kt[RES ? j : 0] = -sdiv;
        const T arg = t_fma(P.tauT, sdiv, xin[j]);
        if (GFN == PROST_FN_SQUARE) xn[j] = t_fma(arg - bc[j], P.rD, bc[j]);
        else { const T v = arg - bc[j]; xn[j] = (v - t_max(t_min(v, P.step), -P.step)) + bc[j]; }
      }
    }
    if constexpr (EXACT) {          // ElemOperation1D<F>, scalar a = 1, d = e = 0: F_prox(v - b; step) + b (Function1DSquare: the exact division)
      T r[VEC];
      if (GFN == PROST_FN_SQUARE) div_to_float_exact_vec<VEC>(parg, P.sq, r);
      else {
#pragma unroll
        for (int j = 0; j < VEC; j++) r[j] = f1d_apply<T, GFN>(GFN, parg[EXACT ? j : 0], P.step, a.g_val[5], a.g_val[6]);
      }
#pragma unroll
      for (int j = 0; j < VEC; j++) xn[j] = r[j] + bc[j];
    }
  };
  // dual step of level l at column c: y^l = prox_f*(y^(l-1) + sigma S K ((1 + theta) x^l - theta x^(l-1)))
  auto dual = [&](auto inner, idx_t c, const LevelStep& P, const T (&xn_c)[VEC], const T (&xn_n)[VEC], const T (&xo_c)[VEC], const T (&xo_n)[VEC],
                  const T (&y1c)[VEC], const T (&y2c)[VEC], T (&o1)[VEC], T (&o2)[VEC], bool acc) {
    constexpr bool I = decltype(inner)::value;
    const bool has_next = I || c + 1 < nx;
    const T bel_n = lane_down(xn_c[0]);            // lane 63: no source, its last row is halo
    const T bel_o = lane_down(xo_c[0]);
    T spd = 0, spv = 0;
    if constexpr (EXACT) {
      // backend_pdhg.cu:341-370 as kernels_fused_iter2.hip evaluates it; ElemOperationNorm2<Function1DIndLeq0> through norm2_leq0_fast
      T av[2][VEC], nv[VEC], out[2][VEC];
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const idx_t row = row0 + j;
        const T below_n = (j < VEC - 1) ? xn_c[(j + 1) % VEC] : bel_n;
        const T below_o = (j < VEC - 1) ? xo_c[(j + 1) % VEC] : bel_o;
        const T kx1 = has_next ? xn_n[j] - xn_c[j] : (T)0;
        const T kx2 = (I || row < ny - 1) ? below_n - xn_c[j] : (T)0;
        const T kp1 = has_next ? xo_n[j] - xo_c[j] : (T)0;
        const T kp2 = (I || row < ny - 1) ? below_o - xo_c[j] : (T)0;
        const T arg1 = y1c[j] + P.sigS * (P.opt * kx1 - P.theta * kp1);
        const T arg2 = y2c[j] + P.sigS * (P.opt * kx2 - P.theta * kp2);
        T norm = 0;
        norm += arg1 * arg1;
        norm += arg2 * arg2;
        av[0][j] = arg1; av[1][j] = arg2; nv[j] = norm;
      }
      norm2_leq0_fast<T, 2, VEC>(nv, av, bq, tiny_is_zero, out);
#pragma unroll
      for (int j = 0; j < VEC; j++) { o1[j] = out[0][j]; o2[j] = out[1][j]; }
      if (RES && acc) {             // the tolerance-compared sums: fp32 with fused multiply-adds, as in the pair kernel's straight-line instances
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          const idx_t row = row0 + j;
          const T below_n = (j < VEC - 1) ? xn_c[(j + 1) % VEC] : bel_n;
          const T kx1 = has_next ? xn_n[j] - xn_c[j] : (T)0;
          const T kx2 = (I || row < ny - 1) ? below_n - xn_c[j] : (T)0;
          const T z1 = (av[0][j] - o1[j]) * inv_sigS, z2 = (av[1][j] - o2[j]) * inv_sigS;
          const T d1 = t_fma(-sqS, kx1, z1), d2 = t_fma(-sqS, kx2, z2);
          spd = t_fma(d1, d1, spd); spd = t_fma(d2, d2, spd); spv = t_fma(z1, z1, spv); spv = t_fma(z2, z2, spv);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < (EXACT ? 0 : VEC); j++) {
      const idx_t row = row0 + j;
      const T below_n = (j < VEC - 1) ? xn_c[(j + 1) % VEC] : bel_n;
      const T below_o = (j < VEC - 1) ? xo_c[(j + 1) % VEC] : bel_o;
      const T kx1 = has_next ? xn_n[j] - xn_c[j] : (T)0;
      const T kx2 = (I || row < ny - 1) ? below_n - xn_c[j] : (T)0;
      const T kp1 = has_next ? xo_n[j] - xo_c[j] : (T)0;
      const T kp2 = (I || row < ny - 1) ? below_o - xo_c[j] : (T)0;
      const T arg1 = t_fma(P.sigS, t_fma(P.opt, kx1, -(P.theta * kp1)), y1c[j]);
      const T arg2 = t_fma(P.sigS, t_fma(P.opt, kx2, -(P.theta * kp2)), y2c[j]);
      // (round 6, timing experiment: with the projection's five instructions per pixel removed -- a fifth of the arithmetic -- a launch of
      // four iterations takes 0.1070 ms against 0.1073: the launch is bound by its 565 MB of traffic at ~5.3 TB/s, not by instruction issue)
      const T sc = t_min(bq * t_rsq(t_fma(arg2, arg2, arg1 * arg1)), (T)1);
      o1[j] = arg1 * sc; o2[j] = arg2 * sc;
      if (RES && acc) {          // primal_residual_transform (backend_pdhg.cu:97-120), see kernels_fused_iter2.hip
        const T z1 = (arg1 - o1[j]) * inv_sigS, z2 = (arg2 - o2[j]) * inv_sigS;
        const T d1 = t_fma(-sqS, kx1, z1), d2 = t_fma(-sqS, kx2, z2);
        spd = t_fma(d1, d1, spd); spd = t_fma(d2, d2, spd); spv = t_fma(z1, z1, spv); spv = t_fma(z2, z2, spv);
      }
    }
    if (RES && acc && owner && c >= rx0 && c < rx1) { r_pd += (double)spd; r_pv += (double)spv; }
  };

  // ---- prologue: zero the ring (lanes outside the image never receive data), start the first R columns ------------------
  {
    V4* rz = reinterpret_cast<V4*>(ring_mem);
    const V4 zero = {};
#pragma unroll
    for (int k = 0; k < R * NB; k++) rz[k * kWave + lane] = zero;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const idx_t c0 = xa - (2 * K - 1);               // first step: P_1 at column c0 + K = xa - K + 1
  if (active && xa - K >= 0) ldv_o<T, VEC, false>(y, off_of(xa - K), Y[0][1][0], VEC);      // y_1^0 at column xa - K: the left neighbour of P_1's first column
#pragma unroll
  for (int r = 0; r < R - 1; r++) if (has_col(c0 + K + r)) ring_issue(c0 + K + r);
  prefetch(c0 + K, c0 - 1);
  ring_read_done();
  // every lane active and no lane on the first / last image row: the whole strip is interior
  const bool strip_inner = (idx_t)strip * kRowsPerWave - H * VEC >= 1 && (idx_t)strip * kRowsPerWave + (idx_t)(kWave - H) * VEC < ny - 1;

  auto step = [&](auto inner, idx_t c) {
    // shift the pipeline by one column
#pragma unroll
    for (int l = 0; l <= K; l++)
#pragma unroll
      for (int j = 0; j < VEC; j++) { X[l][0][j] = X[l][1][j]; X[l][1][j] = X[l][2][j]; }
#pragma unroll
    for (int l = 0; l < K; l++)
#pragma unroll
      for (int j = 0; j < VEC; j++) { Y[l][0][0][j] = Y[l][1][0][j]; Y[l][0][1][j] = Y[l][1][1][j]; }
#pragma unroll
    for (int i = K - 1; i > 0; i--)
#pragma unroll
      for (int j = 0; j < VEC; j++) B[i][j] = B[i - 1][j];
    if (RES) {
#pragma unroll
      for (int j = 0; j < VEC; j++) KTp[RES ? j : 0] = KTn[RES ? j : 0];
    }
    // level 0: column c + K from the register stage
#pragma unroll
    for (int j = 0; j < VEC; j++) { Y[0][1][0][j] = n0[j]; Y[0][1][1][j] = n1[j]; X[0][2][j] = n2[j]; B[0][j] = BPTR ? n3[j] : bval; }
    static_for<1, K + 1>([&](auto L) {
      constexpr int l = decltype(L)::value;
      const LevelStep& P = sp.s[l - 1];
      const idx_t p = c + K - l + 1, q = c + K - l;
      constexpr int xs = l == 1 ? 2 : 1;           // slot of x^(l-1) at column p (level 0 holds one column less)
      // P_l
      if (p >= 0 && p >= xa - (K - l) && p < nx)
        primal(inner, p, P, Y[l - 1][1][0], Y[l - 1][1][1], Y[l - 1][0][0], X[l - 1][xs], B[l - 1], X[l][2], l == K ? KTc : KTn, l >= K - 1);
      if (l == K && owner && p >= xa && p < xb) {
        stv_o<T, VEC, ITERK_NT, false>(x_out, off_of(p), X[K][2], VEC);
        if (RES && p >= rx0 && p < rx1) {          // dual_residual_transform (backend_pdhg.cu:73-94)
          T sdd = 0, sdv = 0;
#pragma unroll
          for (int j = 0; j < VEC; j++) {
            const T w_hat = t_fma(-sqT, KTp[RES ? j : 0], (X[K - 1][1][j] - X[K][2][j]) * inv_tauT);
            const T diff = t_fma(sqT, KTc[RES ? j : 0], w_hat);
            sdd = t_fma(diff, diff, sdd); sdv = t_fma(w_hat, w_hat, sdv);
          }
          r_dd += (double)sdd; r_dv += (double)sdv;
        }
      }
      if constexpr (l == K) prefetch(c + K + 1, c);          // behind the x store, in front of the last dual step and the y stores
      // D_l
      if (q >= 0 && q >= xa - (K - l) && q < nx) {
        if constexpr (l < K) {
          dual(inner, q, P, X[l][1], X[l][2], X[l - 1][xs - 1], X[l - 1][xs], Y[l - 1][0][0], Y[l - 1][0][1], Y[l][1][0], Y[l][1][1], false);
        } else {
          T o1[VEC], o2[VEC];
          dual(inner, q, P, X[l][1], X[l][2], X[l - 1][xs - 1], X[l - 1][xs], Y[l - 1][0][0], Y[l - 1][0][1], o1, o2, true);
          if (owner && q >= xa) { stv_o<T, VEC, ITERK_NT, false>(y_out, off_of(q), o1, VEC); stv_o<T, VEC, ITERK_NT, false>(y2out, off_of(q), o2, VEC); }
        }
      }
    });
    ring_read_done();          // the ds_reads of `prefetch` had the last dual step to complete (no pending read crosses the loop edge)
  };
  for (idx_t c = c0; c < xb; c++) {
    // every stencil of every stage that runs in this step strictly inside the image: the stages of a warm-up step (c < xa) sit at columns
    // >= xa - K + 1 and look one column to the left, so chunks that start at column K or beyond take the select-free instance there too
    if (strip_inner && (c >= xa ? c >= 1 : xa >= K) && c + K + 1 < nx - 1) step(std::true_type(), c);
    else step(std::false_type(), c);
  }
  if (RES) {
    r_pd = wave_sum(r_pd); r_pv = wave_sum(r_pv); r_dd = wave_sum(r_dd); r_dv = wave_sum(r_dv);
    if (lane == 0) {
      double* pp = partial + 4 * (size_t)blockIdx.x;
      pp[0] = r_pd; pp[1] = r_pv; pp[2] = r_dd; pp[3] = r_dv;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
static bool iterk_shape_ok(const prost_hip_fused_desc* d, int dtype) {
  if (!d || dtype != 0 || (d->arith != PROST_HIP_ARITH_FMAD && d->arith != PROST_HIP_ARITH_EXACT)) return false;
  if (d->is3d || d->L != 1 || d->var_T || d->f_moreau || d->g_b_masked) return false;
  if (d->nx < 8 || d->ny < 8 || d->ny % 4 != 0) return false;
  if ((double)d->nx * (double)d->ny * 4 >= 4294967296.0) return false;
  if ((d->g_fn != PROST_FN_SQUARE && d->g_fn != PROST_FN_ABS) || d->f_fn != PROST_FN_IND_LEQ0) return false;
  for (int k = 0; k < 7; k++) {
    if (d->f_coeff_ptr[k]) return false;
    if (d->g_coeff_ptr[k] && k != 1) return false;
  }
  if (!aligned16(d->g_coeff_ptr[1])) return false;
  if ((d->ny + 56 * 4 - 1) / (56 * 4) > (size_t)kReduceBlocks / 2) return false;
  return d->g_coeff_val[0] == 1.0 && d->g_coeff_val[2] != 0.0 && d->g_coeff_val[3] == 0.0 && d->g_coeff_val[4] == 0.0 &&
         d->f_coeff_val[0] == 1.0 && (d->f_coeff_val[1] > 0.0 || d->arith == PROST_HIP_ARITH_EXACT) && d->f_coeff_val[3] == 0.0 && d->f_coeff_val[4] == 0.0;
}

// ring slots and resident wavefronts per SIMD of the instances (160 KB of LDS per CU: 4 SIMDs x WAVES x R x NB KB must fit)
template <int K> struct IterKGeom;
template <> struct IterKGeom<1> { static constexpr int R = 2, W = 4; };      // (one iteration: rebuilding the iterate in front of the last one of a launch)
template <> struct IterKGeom<2> { static constexpr int R = 2, W = 4; };
template <> struct IterKGeom<3> { static constexpr int R = 3, W = 3; };
template <> struct IterKGeom<4> { static constexpr int R = 4, W = 2; };
template <> struct IterKGeom<5> { static constexpr int R = 4, W = 2; };
template <> struct IterKGeom<6> { static constexpr int R = 4, W = 2; };
constexpr int kIterKMax = 6;
constexpr int kIterKMaxExact = 4;      // exact class: instances up to K = 4 (register budget of the correctly rounded forms)
static int iterk_rows_per_wave(int K) { return (kWave - 2 * ((K + 3) / 4)) * 4; }
static int iterk_waves(int K) { return K <= 2 ? IterKGeom<2>::W : K == 3 ? IterKGeom<3>::W : 2; }

static int iterk_override(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && atoi(e) > 0 ? atoi(e) : dflt;
}

static int iterk_chunk_cols(const prost_hip_fused_desc* d, int K, bool res, int cols) {
  const size_t rpw = (size_t)iterk_rows_per_wave(K), strips = (d->ny + rpw - 1) / rpw;
  if (cols <= 0) {
    // The shortest chunk whose workgroups all run in ONE round with two wavefronts per SIMD (2048 on the chip).  Shorter chunks mean a
    // second, mostly empty round (3072^2, K = 4: 2223 workgroups of 18 columns 0.080 ms, 1664 of 24 columns 0.067 ms; 2048^2: 2052 of
    // 9 columns 0.045, 1539 of 12 columns 0.042); longer ones leave SIMDs idle and load 2K - 2 extra columns for less (4096^2, K = 4:
    // 36 columns 0.104, 42 columns 0.112, 72 columns 0.170 ms).  The K = 2 / 3 instances could hold 4 / 3 waves per SIMD, but two already
    // cover the latency: 4096^2, K = 2 with the sums: 36 columns 0.101-0.127 ms between boxes, 18 columns 0.112-0.140.  Images past
    // 2048 workgroups of 72 columns (8192^2) run several rounds whatever the length: 72 (K = 2: 0.457 against 0.503 ms at 36).
    static const int fill = iterk_override("PROST_ITERK_FILL_WAVES", 2);
    const size_t slots = 256 * 4 * (size_t)(iterk_waves(K) < fill ? iterk_waves(K) : fill);
    cols = 72;
    for (int c : {2, 3, 4, 6, 9, 12, 18, 24, 30, 36, 42, 48, 60, 72}) if (strips * ((d->nx + c - 1) / c) <= slots) { cols = c; break; }
    static const char* const names[] = {"", "PROST_ITERK1_COLS", "PROST_ITERK2_COLS", "PROST_ITERK3_COLS", "PROST_ITERK4_COLS", "PROST_ITERK5_COLS", "PROST_ITERK6_COLS"};
    cols = iterk_override(names[K], cols);
  }
  while (res && (size_t)cols < d->nx && strips * ((d->nx + cols - 1) / cols) > (size_t)kReduceBlocks / 2) cols += 6;
  return cols;
}

template <int K>
static int run_iterk(const prost_hip_fused_desc* d, float* x_out, float* y_out, const float* x, const float* y, const double* tau, const double* sigma,
                     const double* theta, int cols, double* out4, void* ws, void* stream, void* record, const RuleTail* tail) {
  if (!iterk_shape_ok(d, 0) || !aligned16(x_out) || !aligned16(y_out) || !aligned16(x) || !aligned16(y)) {
    set_error("fused K-iteration launch: unsupported description"); return 1;
  }
  if (out4 && !ws) { set_error("fused K-iteration launch: residuals need the reduction workspace"); return 1; }
  if (out4 && K < 2) { set_error("fused K-iteration launch: the residual sums need the iterate in front of the last iteration (K >= 2)"); return 1; }
  FusedArgs<float> a = make_fused_args<float>(d);
  const size_t rpw = (size_t)iterk_rows_per_wave(K), strips = (d->ny + rpw - 1) / rpw;
  cols = iterk_chunk_cols(d, K, out4 != nullptr, cols);
  a.cols_per_block = cols;
  a.chunks = (unsigned)((d->nx + cols - 1) / cols);
  if (out4 && strips * a.chunks > (size_t)kReduceBlocks / 2) { set_error("fused K-iteration launch: grid exceeds the reduction workspace"); return 1; }
  if (strips * a.chunks > 0x7fffffffull) { set_error("fused K-iteration launch: grid too large"); return 1; }
  static const int tile_order = iterk_override("PROST_ITERK_TILE_ORDER", 0);
  a.strips = tile_order == 1 ? (unsigned)strips : 0u;
  StepsK<K> sp;
  for (int l = 0; l < K; l++) {
    const float t = record ? 1.0f : (float)tau[l], s = record ? 1.0f : (float)sigma[l], th = record ? 1.0f : (float)theta[l];
    const UniformProx<float> ug = make_uniform_prox<float>(a.g_val, t * a.Tval);
    if (!record && !(ug.a_one && ug.den_one && !ug.degenerate)) { set_error("fused K-iteration launch: not the straight-line shape"); return 1; }
    sp.s[l].tauT = t * a.Tval; sp.s[l].rD = (float)ug.sq.rD; sp.s[l].step = ug.step; sp.s[l].sigS = s * a.Sval; sp.s[l].theta = th; sp.s[l].opt = 1 + th;
    sp.s[l].sq = ug.sq;
  }
  sp.tau_last = record ? 1.0f : (float)tau[K - 1]; sp.sigma_last = record ? 1.0f : (float)sigma[K - 1];
  dim3 grid((unsigned)(strips * a.chunks)), block(kWave);
  hipStream_t s = as_stream(stream);
  double* partial = static_cast<double*>(ws);
  const PdhgRecord<float>* rec = static_cast<const PdhgRecord<float>*>(record);
  constexpr int R = IterKGeom<K>::R, W = IterKGeom<K>::W;
  static const int geom = iterk_override("PROST_ITERK_GEOM", 0);
  const bool exact = d->arith == PROST_HIP_ARITH_EXACT;
#define GOKA(G, BP, RS, EX) PH_LAUNCH((fused_iter2d_xk_kernel<K, G, BP, RS, R, (RS && W > 2 ? W - 1 : W), EX>), grid, block, 0, s, x_out, y_out, x, y, a, sp, partial, rec)
#define GOK(G, BP, RS) do { if (exact) { if constexpr (K <= kIterKMaxExact) GOKA(G, BP, RS, true); else { set_error("fused K-iteration launch: the exact class runs K <= 4"); return 1; } } \
    else GOKA(G, BP, RS, false); } while (0)
#define GOKW(G, BP, RS) PH_LAUNCH((fused_iter2d_xk_kernel<K, G, BP, RS, 8, 1, false>), grid, block, 0, s, x_out, y_out, x, y, a, sp, partial, rec)
#define GOK2(G, BP) do { \
    bool done = false; \
    if constexpr (K == 4 && G == PROST_FN_SQUARE && BP) { if (geom == 1) { done = true; if (out4) GOKW(G, BP, true); else GOKW(G, BP, false); } } \
    if (!done) { if (out4) GOK(G, BP, true); else GOK(G, BP, false); } \
  } while (0)
  const bool bp = d->g_coeff_ptr[1] != nullptr;
  if (d->g_fn == PROST_FN_SQUARE) { if (bp) GOK2(PROST_FN_SQUARE, true); else GOK2(PROST_FN_SQUARE, false); }
  else { if (bp) GOK2(PROST_FN_ABS, true); else GOK2(PROST_FN_ABS, false); }
#undef GOK2
#undef GOKW
#undef GOK
#undef GOKA
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "fused K-iteration kernel"); }
  if (out4 && tail && tail->apply) return launch_fold4_rule<float>(out4, partial, grid.x, record, tail->iteration, tail->mirror, s);
  if (out4) return launch_fold4(out4, partial, grid.x, s);
  return 0;
}

static int run_iterk_any(int K, const prost_hip_fused_desc* d, float* x_out, float* y_out, const float* x, const float* y, const double* tau,
                         const double* sigma, const double* theta, int cols, double* out4, void* ws, void* stream, void* record, const RuleTail* tail) {
  switch (K) {
    case 1: return run_iterk<1>(d, x_out, y_out, x, y, tau, sigma, theta, cols, out4, ws, stream, record, tail);
    case 2: return run_iterk<2>(d, x_out, y_out, x, y, tau, sigma, theta, cols, out4, ws, stream, record, tail);
    case 3: return run_iterk<3>(d, x_out, y_out, x, y, tau, sigma, theta, cols, out4, ws, stream, record, tail);
    case 4: return run_iterk<4>(d, x_out, y_out, x, y, tau, sigma, theta, cols, out4, ws, stream, record, tail);
    case 5: return run_iterk<5>(d, x_out, y_out, x, y, tau, sigma, theta, cols, out4, ws, stream, record, tail);
    case 6: return run_iterk<6>(d, x_out, y_out, x, y, tau, sigma, theta, cols, out4, ws, stream, record, tail);
  }
  set_error("fused K-iteration launch: K must be 1 .. 6");
  return 1;
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_fused_iterationk_max(const prost_hip_fused_desc* desc, int dtype) {
  return !iterk_shape_ok(desc, dtype) ? 0 : desc->arith == PROST_HIP_ARITH_EXACT ? kIterKMaxExact : kIterKMax;
}
int prost_hip_fused_iterationk_chunk_cols(const prost_hip_fused_desc* desc, int dtype, int k, int with_residuals) {
  return iterk_shape_ok(desc, dtype) && k >= 1 && k <= (desc->arith == PROST_HIP_ARITH_EXACT ? kIterKMaxExact : kIterKMax) ? iterk_chunk_cols(desc, k, with_residuals != 0, 0) : 0;
}
int prost_hip_fused_iterationk_f32(const prost_hip_fused_desc* d, int k, float* x_out, float* y_out, const float* x, const float* y, const double* tau,
                                   const double* sigma, const double* theta, int cols_per_block, double* res_out4, void* workspace, void* s) {
  return run_iterk_any(k, d, x_out, y_out, x, y, tau, sigma, theta, cols_per_block, res_out4, workspace, s, nullptr, nullptr);
}
int prost_hip_fused_iterationk_rec_f32(const prost_hip_fused_desc* d, int k, float* x_out, float* y_out, const float* x, const float* y, void* record,
                                       int cols_per_block, double* res_out4, void* workspace, int apply_rule, unsigned long long iteration,
                                       prost_hip_pdhg_rule_state* mirror, void* s) {
  if (!record) { set_error("fused K-iteration launch: no step-size record"); return 1; }
  const RuleTail tail = {apply_rule, iteration, mirror};
  return run_iterk_any(k, d, x_out, y_out, x, y, nullptr, nullptr, nullptr, cols_per_block, res_out4, workspace, s, record, &tail);
}
// (fp64 has no tolerance-class instances: prost_hip_fused_iterationk_max answers 0 for it)
int prost_hip_fused_iterationk_f64(const prost_hip_fused_desc*, int, double*, double*, const double*, const double*, const double*, const double*, const double*, int,
                                   double*, void*, void*) {
  set_error("fused K-iteration launch: fp32 only"); return 1;
}
int prost_hip_fused_iterationk_rec_f64(const prost_hip_fused_desc*, int, double*, double*, const double*, const double*, void*, int, double*, void*, int,
                                       unsigned long long, prost_hip_pdhg_rule_state*, void*) {
  set_error("fused K-iteration launch: fp32 only"); return 1;
}
}  // extern "C"
