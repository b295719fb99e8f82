// kernels_fused_iter2.hip -- TWO PDHG iterations per kernel launch (temporal blocking) for gradient2d.
//
// The single-iteration kernel (kernels_fused_iter.hip) is HBM-bound at 7 floats/pixel/iteration.
// The only way below that is to keep x^(k+1), y^(k+1) on chip: this kernel reads x^k, y^k and the
// prox_g coefficient vectors once and writes x^(k+2), y^(k+2) -- 7 floats/pixel per TWO iterations.
// A wavefront marches over a chunk of columns with a 4-stage software pipeline, all in registers:
//     A(c+2): x1 = primal step of iteration k    at column c+2   (needs y^k    at c+2, c+1)
//     B(c+1): y1 = dual   step of iteration k    at column c+1   (needs x1     at c+1, c+2)
//     C(c+1): x2 = primal step of iteration k+1  at column c+1   (needs y1     at c+1, c)
//     D(c)  : y2 = dual   step of iteration k+1  at column c     (needs x2     at c,   c+1)
// Row neighbours come from adjacent lanes (64-lane shuffles).  Lane 0 and lane 63 are HALO lanes:
// with R the wave's first row, stage A is wrong at row R (no row above), B at R and the last row,
// C at R, R+1 and the last row, D at R..R+1 and the last two rows -- all inside the halo lanes'
// VEC rows, which are never stored.  The wave therefore owns 62 x VEC rows; neighbouring strips
// overlap by two lanes.  Column halo: a
// chunk starts its pipeline 3 columns early (A(xa-1), B(xa-1) are recomputed) and runs A, B, C one
// resp. two columns past its end.
// Every stage evaluates exactly the expressions of the single-iteration kernel, so x^(k+2), y^(k+2)
// are bit-identical to two single launches.  Intermediate iterates are NOT stored: the host only
// pairs iterations whose intermediate state nobody reads (see BackendPDHG::PerformIterations).
#include "fused_common.hpp"
#include "reduce.hpp"

#include <type_traits>

namespace prost_hip {

typedef int idx_t;            // column / row index inside the kernels
constexpr int popcount7b(int m) { int c = 0; for (int k = 0; k < 7; k++) c += (m >> k) & 1; return c; }
constexpr int slot_ofb(int m, int k) { int c = 0; for (int i = 0; i < k; i++) c += (m >> i) & 1; return c; }

// PF == 0 selects the LDS prefetch ring (see the kernel): resident wavefronts per SIMD of that instance
constexpr int kRingWaves = 4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <class T, int VEC, int GMASK>
struct Col2 {
  static constexpr int NG = popcount7b(GMASK) > 0 ? popcount7b(GMASK) : 1;
  T y1[VEC], y2[VEC], x[VEC], gc[NG][VEC];
};

// MODE bit 0: additionally store the intermediate iterate x^(k+1), y^(k+1) (x_mid, y_mid; 10 floats/pixel
// instead of 7).  MODE bit 1: the four residual sums of iteration k+1 (backend_pdhg.cu:392-431) --
// everything they need (y^k, y^(k+1), y^(k+2), x^(k+1), x^(k+2), K^T y^k, K^T y^(k+1), K x^(k+1),
// K x^(k+2)) is in registers, no extra HBM traffic.
// PF: columns of loads kept in flight per wave (3 for the straight-line instances at 3 waves/SIMD, 1 otherwise)
// VART: position-dependent primal preconditioner (FusedArgs::varT; see fused_iter2d_kernel): the interior instance of a column step
// (`inner`) is untouched -- every pixel it sees has 4 stencil entries in its column, Tval = Tcls[2] -- and the boundary instance
// re-evaluates the pixels of the first / last column and row with their own Tau_j through the reference's expression (elem_1d).
// FMAD: the tolerance-class arithmetic (prost_hip_fused_desc.arith = PROST_HIP_ARITH_FMAD; straight-line ROF / TV-L1 shapes only): the
// SAME expressions with fused multiply-adds where a product feeds a sum (what nvcc's default -fmad=true makes of the reference's
// kernels, src/CMakeLists.txt:12-24), the quotient by the wave-uniform 1 + step as a product with its fp32 reciprocal and
// pr v / ||v|| as v * min(b * rsq(||v||^2), 1) (v_rsq_f32, 1 ulp) -- no fp64 instruction, no conversion, no range guard.  Results are
// within a stated tolerance of the exact instances (tests/test_gpu_fmad.py), not bit-identical to them.
template <class T, int VEC, int GFN, int FFN, int GMASK, int PF, bool FAST, int MODE, bool RAG, bool VART, bool FMAD>
__global__ void __launch_bounds__(kWave, FAST ? (PF == 0 ? ((MODE & 2) ? 3 : kRingWaves) : (MODE & 2) ? 2 : (PF > 1 || MODE == 1 ? 3 : 4)) : 1)
    fused_iter2d_x2_kernel(T* __restrict__ x_out, T* __restrict__ y_out,
                                                                const T* __restrict__ x, const T* __restrict__ y,
                                                                T* __restrict__ x_mid, T* __restrict__ y_mid,
                                                                FusedArgs<T> a, IterParams<T> p1, IterParams<T> p2,
                                                                double* __restrict__ partial, const PdhgRecord<T>* __restrict__ rec) {
  if (rec) {                       // device-resident step sizes (fused_common.hpp: PdhgRecord): wave-uniform scalar loads
    if (rec->stop) return;
    p1 = rec->p; p2 = rec->p;      // a rule evaluation never falls between the two iterations of a launch
  }
  // 32-bit indices: columns and rows are < 2^31 (iter2_desc_ok: one plane is < 4 GiB), and the scalar unit compares 32-bit values
  // itself -- 64-bit orderings are vector instructions
  const idx_t nx = (idx_t)a.nx, ny = (idx_t)a.ny;
  const idx_t rx0 = (idx_t)a.rx0, rx1 = a.rx1 > (size_t)0x7fffffff ? (idx_t)0x7fffffff : (idx_t)a.rx1;
  const int lane = threadIdx.x;
  constexpr int kRowsPerWave = (kWave - 2) * VEC;
  const unsigned total = gridDim.x, chunks = a.chunks;
  const unsigned xcd = blockIdx.x % 8u, q = blockIdx.x / 8u;          // XCD-aware tile order (kernels_fused_iter.hip)
  const unsigned tile = xcd * (total / 8u) + (xcd < total % 8u ? xcd : total % 8u) + q;
  // (tried in round 3: vertically adjacent strips as consecutive tiles -- they march over the same columns at the same time, so the
  // rows their halo lanes share and the cache lines their boundaries straddle (a strip is 248 floats = 7.75 lines per column) meet
  // in the XCD's L2.  FETCH_SIZE per launch drops from 177.0 to 155.4 thousand KiB (groups of 2 / 4 strips: 169.6 / 164.9), the
  // total traffic from 564 to 522 MB -- and the launch gets SLOWER, 0.108 against 0.105 ms on the same box: at 5.4 TB/s the kernel
  // is bound by what a wavefront has in flight, not by the bytes; the chunks of one strip stay consecutive.)
  const unsigned strip = tile / chunks, chunk = tile % chunks;
  const idx_t row0 = (idx_t)strip * kRowsPerWave + ((idx_t)lane - 1) * VEC;
  const bool active = row0 >= 0 && row0 < ny;
  const bool owner = active && lane > 0 && lane < kWave - 1;
  // RAG: the image height is not a multiple of VEC (fused_common.hpp, ldv_n / stv_n)
  const int nvalid = !RAG ? VEC : (active ? (ny - row0 < (idx_t)VEC ? (int)(ny - row0) : VEC) : 0);   // rows of this lane inside the image
  const idx_t xa = (idx_t)chunk * a.cols_per_block;
  const idx_t xb = xa + a.cols_per_block < nx ? xa + a.cols_per_block : nx;
  const size_t N = (size_t)nx * (size_t)ny;
  constexpr bool kUniformG = (GMASK & 0x15) == 0;
  constexpr bool kBMask = FAST && (GMASK & 0x80) != 0;     // bit 7: the per-pixel b carries the mask sentinel (binary coefficient a folded in)
  constexpr bool kRes = (MODE & 2) != 0;
  constexpr bool kMid = (MODE & 1) != 0;
  typedef Col2<T, VEC, GMASK> Col;
  const T sqT = t_sqrt(a.Tval), sqS = t_sqrt(a.Sval);
  const bool tiny_is_zero = a.f_val[1] >= (T)kTinyIsZeroRadius;
  // straight-line instance whose prox_f* is the Moreau wrap of ElemOperationNorm2<FFN> (FFN = abs: ROF written in its primal form,
  // example_rof_primal.m:27; kernels_fused_iter.hip, device_math.hpp: norm2_moreau_post)
  constexpr bool kFM = FAST && FFN != PROST_FN_IND_LEQ0;
  double r_pd = 0, r_pv = 0, r_dd = 0, r_dv = 0;       // primal diff^2, primal var^2, dual diff^2, dual var^2

  // one 32-bit byte offset per lane serves every plane (x, y1, y2, coefficients, outputs): the plane
  // bases are wave-uniform (SGPR pairs), so the loads/stores use the saddr + 32-bit voffset form and
  // no 64-bit per-lane address arithmetic is left in the loop (host guarantees N * sizeof(T) < 4 GiB)
  auto off_of = [&](idx_t c) { return ((unsigned)c * (unsigned)ny + (unsigned)row0) * (unsigned)sizeof(T); };   // mod 2^32: exact for every (column, row) inside the image
  // ---- PF == 0: prefetch through an LDS ring instead of a register ring --------------------------------------------------
  // The register version keeps the columns c+1 .. c+2+PF of the four input streams in VGPRs (80 registers at PF = 3) and
  // moves them one slot per step (64 v_mov per column).  Here a column is fetched by four LDS-DMA loads
  // (global_load_lds_dwordx4: no VGPR, wave-uniform LDS base + lane * 16) two steps before it is used, lands in one of two
  // 4-KiB slots of the wave's own LDS ring and is read into the `in2` registers when its step comes; only in1 <- in2 is
  // still a register copy.  ~48 VGPRs and ~48 moves per column less, which fits four resident wavefronts per SIMD.
  // Nothing orders an LDS read behind a pending LDS-DMA except the issuing wave's vmcnt, and hipcc drains vmcnt(0) before
  // any LDS read it can see, so the ring is read with inline-asm ds_read_b128 behind a COUNTED s_waitcnt: at the top of a
  // step the batch of column c+3 (NB loads, issued one step ago) may still be in flight, whatever stores lie in between.
  constexpr bool kRing = PF == 0;
  constexpr int NB = 3 + ((GMASK >> 1) & 1);                    // LDS-DMA loads per column: y1, y2, x (, b of prox_g)
  __shared__ __attribute__((aligned(16))) char ring_mem[kRing ? 2 * 4096 : 16];
  const T* const y2base = y + N;
  T* const y2out = y_out + N;
  T* const y2mid = kMid ? y_mid + N : nullptr;
  auto load_col = [&](idx_t c, Col& in) {
    const unsigned o = off_of(c);
    ldv_o<T, VEC, RAG>(y, o, in.y1, nvalid); ldv_o<T, VEC, RAG>(y2base, o, in.y2, nvalid); ldv_o<T, VEC, RAG>(x, o, in.x, nvalid);
#pragma unroll
    for (int k = 0; k < 7; k++) {
      if ((GMASK >> k) & 1) {
        if (a.g_ptr[k]) ldv_o<T, VEC, RAG>(a.g_ptr[k], o, in.gc[slot_ofb(GMASK, k)], nvalid);
        else {
#pragma unroll
          for (int j = 0; j < VEC; j++) in.gc[slot_ofb(GMASK, k)][j] = a.g_val[k];
        }
      }
    }
  };
  auto has_col = [&](idx_t k) { return k >= 0 && k < nx && k <= xb + 1; };
  auto ring_issue = [&](idx_t k) {                                // column k -> slot k & 1
    if (active) {
      const unsigned o = off_of(k);
      char* slot = ring_mem + (k & 1) * 4096;
      __builtin_amdgcn_global_load_lds((glb_void_t*)(reinterpret_cast<const char*>(y) + o), (lds_void_t*)(slot), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(reinterpret_cast<const char*>(y2base) + o), (lds_void_t*)(slot + 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(reinterpret_cast<const char*>(x) + o), (lds_void_t*)(slot + 2048), 16, 0, 0);
      if ((GMASK >> 1) & 1) __builtin_amdgcn_global_load_lds((glb_void_t*)(reinterpret_cast<const char*>(a.g_ptr[1]) + o), (lds_void_t*)(slot + 3072), 16, 0, 0);
    }
  };
  auto ring_fetch = [&](idx_t k, bool next_in_flight, Col& in) {
    typedef typename VecOf<T>::native V4;
    if (next_in_flight) { if (NB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)ring_mem) + (unsigned)(k & 1) * 4096u + (unsigned)lane * 16u;
    V4 v0, v1, v2, v3 = {};
    asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(addr) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(v1) : "v"(addr) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(v2) : "v"(addr) : "memory");
    if ((GMASK >> 1) & 1) asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(v3) : "v"(addr) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) :: "memory");
#pragma unroll
    for (int j = 0; j < VEC; j++) { in.y1[j] = v0[j]; in.y2[j] = v1[j]; in.x[j] = v2[j]; if ((GMASK >> 1) & 1) in.gc[slot_ofb(GMASK, 1)][j] = v3[j]; }
  };
  // primal step at column c (backend_pdhg.cu:317-338 with block_gradient2d.cu:122-138 inlined)
  // `inner` (a compile-time tag): every row of the wave and the columns c-1 .. c+1 are strictly inside
  // the image, so the boundary selects of the gradient stencil drop out (identical values otherwise)
  auto primal = [&](auto inner, idx_t c, const T (&y1c)[VEC], const T (&y2c)[VEC], T up, const T (&y1p)[VEC], const T (&xin)[VEC],
                    const T (&gc)[Col::NG][VEC], const IterParams<T>& P, T (&xn)[VEC], T (&kt)[VEC]) {
    constexpr bool I = decltype(inner)::value;
    const T tauT = P.tau * a.Tval;
    if constexpr (FMAD) {
      // x - tau T K^T y = fma(tau T, div y, x);  Function1DSquare: (v - b) / (1 + step) + b = fma(v - b, 1 / (1 + step), b);
      // Function1DAbs: soft threshold of v - b by step
      const T rD = (T)P.ug.sq.rD, st = P.ug.step;
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const idx_t row = row0 + j;
        const T upj = (j > 0) ? y2c[(j + VEC - 1) % VEC] : up;
        const T divy = ((I || row < ny - 1) ? y2c[j] : (T)0) - ((I || row > 0) ? upj : (T)0);
        const T divx = ((I || c < nx - 1) ? y1c[j] : (T)0) - ((I || c > 0) ? y1p[j] : (T)0);
        const T sdiv = divx + divy;
        kt[j] = -sdiv;
        const T arg = t_fma(tauT, sdiv, xin[j]);
        const T bj = ((GMASK >> 1) & 1) ? gc[slot_ofb(GMASK, 1)][j] : a.g_val[1];
        T r;
        if (GFN == PROST_FN_SQUARE) r = t_fma(arg - bj, rD, bj);
        else { const T v = arg - bj; r = (v - t_max(t_min(v, st), -st)) + bj; }
        if (kBMask) r = is_mask_sentinel(bj) ? arg : r;
        xn[j] = r;
      }
      return;
    }
    constexpr bool kEdges = VART && !I;          // this instance may see pixels with fewer than 4 stencil entries in their column
    T parg[VEC], parg0[kBMask ? VEC : 1];
    T argv[kEdges ? VEC : 1], tTv[kEdges ? VEC : 1];
    bool edgev[kEdges ? VEC : 1], cornerv[kEdges ? VEC : 1];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const idx_t row = row0 + j;
      T divy = (I || row < ny - 1) ? y2c[j] : (T)0;
      if (I || row > 0) divy -= (j > 0) ? y2c[(j + VEC - 1) % VEC] : up;
      T divx = (I || c < nx - 1) ? y1c[j] : (T)0;
      if (I || c > 0) divx -= y1p[j];
      T kty = (T)0 - (divx + divy);
      if (VART) {                // K^T y in the order of the matrix's transposed CSR row (kernels_fused_iter.hip: fused_iter2d_kernel)
        T s = 0;
        if (I || c > 0) s += y1p[j];
        if (I || c < nx - 1) s -= y1c[j];
        if (I || row > 0) s += (j > 0) ? y2c[(j + VEC - 1) % VEC] : up;
        if (I || row < ny - 1) s -= y2c[j];
        kty = s;
      }
      kt[j] = kty;
      T tT = tauT;
      bool edge = false;
      if (kEdges) {
        const int cnt = 4 - (c == 0 ? 1 : 0) - (c == nx - 1 ? 1 : 0) - (row == 0 ? 1 : 0) - (row == ny - 1 ? 1 : 0);
        edge = cnt != 4;
        tT = P.tau * (cnt == 4 ? a.Tval : (cnt == 3 ? a.Tcls[1] : a.Tcls[0]));
        edgev[kEdges ? j : 0] = edge; cornerv[kEdges ? j : 0] = cnt == 2; tTv[kEdges ? j : 0] = tT;
      }
      const T arg = xin[j] - tT * kty;
      if (kEdges) argv[kEdges ? j : 0] = arg;
      if (FAST) {
        if (kBMask) parg0[j] = arg;
        parg[j] = arg - (((GMASK >> 1) & 1) ? gc[slot_ofb(GMASK, 1)][j] : a.g_val[1]);
      } else {
        T cf[7];
#pragma unroll
        for (int k = 0; k < 7; k++) cf[k] = ((GMASK >> k) & 1) ? gc[slot_ofb(GMASK, k)][j] : a.g_val[k];
        if (kUniformG && !edge) xn[j] = elem_1d_u<T, GFN>(a.g_fn, arg, cf, P.ug);
        else xn[j] = elem_1d<T, GFN>(a.g_fn, arg, tT, cf);
      }
    }
    if (FAST) {
      // ElemOperation1D<F> with scalar a = 1, d = 0, e = 0 (host-checked): a (v - d tau) = v and the fp64
      // denominator is 1, so the scaled prox is  F_prox(v - b; step) + b  -- straight-line code for the
      // VEC elements.  F = Function1DSquare: the exact division (one fallback branch per vector);
      // F = Function1DAbs (TV-L1 data term): soft threshold by step = c tau.
      T r[VEC];
      if (GFN == PROST_FN_SQUARE) div_to_float_exact_vec<VEC>(parg, P.ug.sq, r);
      else {
#pragma unroll
        for (int j = 0; j < VEC; j++) r[j] = f1d_apply<T, GFN>(a.g_fn, parg[j], P.ug.step, a.g_val[5], a.g_val[6]);
      }
#pragma unroll
      for (int j = 0; j < VEC; j++) xn[j] = r[j] + (((GMASK >> 1) & 1) ? gc[slot_ofb(GMASK, 1)][j] : a.g_val[1]);
      if (kBMask) {
        // merged b stream (prost_hip_mask_merge): where the coefficient a of prox_g is 0 the element passes through,
        // (arg - tau d) / (1 + tau e) = arg for d = e = 0 (elem_operation_1d.hpp:42-44); elsewhere a = 1, the code above
#pragma unroll
        for (int j = 0; j < VEC; j++) if (is_mask_sentinel(gc[slot_ofb(GMASK, 1)][j])) xn[j] = parg0[kBMask ? j : 0];
      }
      if (kEdges) {
        // pixels with their own Tau_j.  Straight-line square shape: the same F_prox(v - b; step_j) + b with the divisor 1 + step_j of the
        // pixel's class (two classes, their reciprocals formed once per launch: IterParams::ec) -- (float)((double)(v - b) / D_j) exactly
        // as elem_1d evaluates it; any other shape: ElemOperation1D as the reference writes it.
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          if (edgev[kEdges ? j : 0]) {
            if (GFN == PROST_FN_SQUARE && !kBMask) {
              const bool cn = cornerv[kEdges ? j : 0];
              UniformDiv dv;
              dv.D = cn ? P.ec[0].sq.D : P.ec[1].sq.D; dv.rD = cn ? P.ec[0].sq.rD : P.ec[1].sq.rD;
              const T bj = ((GMASK >> 1) & 1) ? gc[slot_ofb(GMASK, 1)][j] : a.g_val[1];
              xn[j] = div_to_float_exact(argv[kEdges ? j : 0] - bj, dv) + bj;
            } else {
              T cf[7];
#pragma unroll
              for (int k = 0; k < 7; k++) cf[k] = ((GMASK >> k) & 1) ? gc[slot_ofb(GMASK, k)][j] : a.g_val[k];
              xn[j] = elem_1d<T, GFN>(a.g_fn, argv[kEdges ? j : 0], tTv[kEdges ? j : 0], cf);
            }
          }
        }
      }
    }
  };
  // primal_residual_transform (backend_pdhg.cu:97-120) of one pixel, both components
  // the residual divisors sigma sqrt(S), tau sqrt(T) are wave-uniform: one double reciprocal each (device_math.hpp)
  const SharedDivisor<T> div_sigS(p2.sigma * sqS), div_tauT(p2.tau * sqT);
  const T inv_sigS = (T)1 / (p2.sigma * sqS), inv_tauT = (T)1 / (p2.tau * sqT);
  auto residual_terms = [&](int j, T y1o, T y2o, T o1, T o2, T kx1, T kx2, T kp1, T kp2, const IterParams<T>& P) {
    (void)j;
    const T z1 = div_sigS.div(y1o - o1) + sqS * ((1 + P.theta) * kx1 - P.theta * kp1);
    const T d1 = z1 - sqS * kx1;
    r_pd += (double)(d1 * d1); r_pv += (double)(z1 * z1);
    const T z2 = div_sigS.div(y2o - o2) + sqS * ((1 + P.theta) * kx2 - P.theta * kp2);
    const T d2 = z2 - sqS * kx2;
    r_pd += (double)(d2 * d2); r_pv += (double)(z2 * z2);
  };
  // dual step at column c (backend_pdhg.cu:341-370 with block_gradient2d.cu:61-77 inlined)
  auto dual = [&](auto inner, idx_t c, const T (&xn_c)[VEC], const T (&xn_n)[VEC], const T (&xo_c)[VEC], const T (&xo_n)[VEC],
                  const T (&y1c)[VEC], const T (&y2c)[VEC], const IterParams<T>& P, T (&o1)[VEC], T (&o2)[VEC], bool acc) {
    constexpr bool I = decltype(inner)::value;
    const T sigS = P.sigma * a.Sval, theta = P.theta;
    const bool has_next = I || c + 1 < nx;
    const bool counted = c >= rx0 && c < rx1;       // residual terms of this column count (column-sharded images)
    const T bel_n = lane_down(xn_c[0]);                         // lane 63: no source, its last row is halo
    const T bel_o = lane_down(xo_c[0]);
    T av[2][VEC], nv[VEC];
    if constexpr (FMAD) {
      const T opt = 1 + theta, bq = a.f_val[1];
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const idx_t row = row0 + j;
        const T below_n = (j < VEC - 1) ? xn_c[(j + 1) % VEC] : bel_n;
        const T below_o = (j < VEC - 1) ? xo_c[(j + 1) % VEC] : bel_o;
        const T kx1 = has_next ? xn_n[j] - xn_c[j] : (T)0;
        const T kx2 = (I || row < ny - 1) ? below_n - xn_c[j] : (T)0;
        const T kp1 = has_next ? xo_n[j] - xo_c[j] : (T)0;
        const T kp2 = (I || row < ny - 1) ? below_o - xo_c[j] : (T)0;
        const T arg1 = t_fma(sigS, t_fma(opt, kx1, -(theta * kp1)), y1c[j]);
        const T arg2 = t_fma(sigS, t_fma(opt, kx2, -(theta * kp2)), y2c[j]);
        // ElemOperationNorm2<Function1DIndLeq0>, a = 1, d = e = 0, radius b > 0: pr v / ||v|| = v min(b / ||v||, 1); ||v|| = 0: b * inf -> 1, v = 0
        const T sc = t_min(bq * t_rsq(t_fma(arg2, arg2, arg1 * arg1)), (T)1);
        av[0][j] = arg1; av[1][j] = arg2;
        o1[j] = arg1 * sc; o2[j] = arg2 * sc;
      }
    }
    T vv[kFM ? 2 : 1][kFM ? VEC : 1];          // kFM: the pre-scaled arguments v = arg / (sigma Sigma)
    const SharedDivisor<T> div_sS(kFM ? sigS : (T)1);
#pragma unroll
    for (int j = 0; j < (FMAD ? 0 : VEC); j++) {
      const idx_t row = row0 + j;
      const T below_n = (j < VEC - 1) ? xn_c[(j + 1) % VEC] : bel_n;
      const T below_o = (j < VEC - 1) ? xo_c[(j + 1) % VEC] : bel_o;
      const T kx1 = has_next ? xn_n[j] - xn_c[j] : (T)0;
      const T kx2 = (I || row < ny - 1) ? below_n - xn_c[j] : (T)0;
      const T kp1 = has_next ? xo_n[j] - xo_c[j] : (T)0;
      const T kp2 = (I || row < ny - 1) ? below_o - xo_c[j] : (T)0;
      const T arg1 = y1c[j] + sigS * ((1 + theta) * kx1 - theta * kp1);
      const T arg2 = y2c[j] + sigS * ((1 + theta) * kx2 - theta * kp2);
      const T w1 = kFM ? div_sS.div(arg1) : arg1, w2 = kFM ? div_sS.div(arg2) : arg2;
      T norm = 0;
      norm += w1 * w1;
      norm += w2 * w2;
      if (FAST) {
        if (kFM) { vv[0][kFM ? j : 0] = w1; vv[kFM ? 1 : 0][kFM ? j : 0] = w2; }
        av[0][j] = arg1; av[1][j] = arg2; nv[j] = norm;
      } else if (norm > 0) {
        norm = t_sqrt(norm);
        const T pr = scaled_prox_u<T, FFN>(a.f_fn, norm, a.f_val, P.uf);
        o1[j] = pr * arg1 / norm;
        o2[j] = pr * arg2 / norm;
      } else {
        o1[j] = 0; o2[j] = 0;
      }
      if (!FAST && kRes && acc && owner && counted && j < nvalid) residual_terms(j, y1c[j], y2c[j], o1[j], o2[j], kx1, kx2, kp1, kp2, P);
    }
    if (FAST) {
      // ElemOperationNorm2<Function1DIndLeq0> with scalar a = 1, d = 0, e = 0 (host-checked): out = pr v / ||v||,
      // pr = min(||v|| - b, 0) + b, 0 for ||v|| = 0 -- straight-line for the VEC pixels (device_math.hpp)
      if constexpr (!FMAD) {
        T out[2][VEC];
        if constexpr (kFM) norm2_moreau_post<T, FFN, kFM ? 2 : 1, kFM ? VEC : 1>(nv, vv, av, sigS, a.f_val, P.uf, out);
        else norm2_leq0_fast<T, 2, VEC>(nv, av, a.f_val[1], tiny_is_zero, out);
#pragma unroll
        for (int j = 0; j < VEC; j++) { o1[j] = out[0][j]; o2[j] = out[1][j]; }
      }
      if (kRes && acc && owner && counted) {
        // primal_residual_transform (backend_pdhg.cu:97-120): z_hat = (y_old - y_new) / (sigma sqrt(S)) + sqrt(S) ((1 + theta) K x_new
        // - theta K x_old).  With the prox argument above, arg = y_old + sigma S ((1 + theta) K x_new - theta K x_old), that is
        // (arg - y_new) / (sigma sqrt(S)): one subtraction and one product with the wave-uniform reciprocal.  The residual SUMS are
        // compared with a relative tolerance (they steer step sizes and the stopping test; the reference reduces them in T in an
        // unspecified order), so this block -- unlike the iterates -- uses plain fp32 with fused multiply-adds, summed per column
        // in T and across columns in double.
        T spd = 0, spv = 0;
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          const idx_t row = row0 + j;
          const T below_n = (j < VEC - 1) ? xn_c[(j + 1) % VEC] : bel_n;
          const T kx1 = has_next ? xn_n[j] - xn_c[j] : (T)0;     // K x^(k+2) again: cheaper than keeping it in registers
          const T kx2 = (I || row < ny - 1) ? below_n - xn_c[j] : (T)0;
          const T z1 = (av[0][j] - o1[j]) * inv_sigS, z2 = (av[1][j] - o2[j]) * inv_sigS;
          const T d1 = t_fma(-sqS, kx1, z1), d2 = t_fma(-sqS, kx2, z2);
          if (j < nvalid) { spd = t_fma(d1, d1, spd); spd = t_fma(d2, d2, spd); spv = t_fma(z1, z1, spv); spv = t_fma(z2, z2, spv); }
        }
        r_pd += (double)spd; r_pv += (double)spv;
      }
    }
  };

  Col in1 = {}, in2 = {};                       // loaded columns c+1 and c+2
  T x1_0[VEC], x1_1[VEC], x1_2[VEC];            // x^(k+1) at columns c, c+1, c+2
  T y1a_0[VEC], y1b_0[VEC], y1a_1[VEC], y1b_1[VEC];   // y^(k+1) (both components) at columns c, c+1
  T x2_0[VEC], x2_1[VEC];                       // x^(k+2) at columns c, c+1
  T kt_1[VEC], kt_2[VEC], kt_c[VEC];            // K^T y^k at columns c+1, c+2; K^T y^(k+1) at c+1 (residuals)
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    x1_0[j] = x1_1[j] = x1_2[j] = 0; y1a_0[j] = y1b_0[j] = y1a_1[j] = y1b_1[j] = 0; x2_0[j] = x2_1[j] = 0;
    kt_1[j] = kt_2[j] = kt_c[j] = 0;
  }
  // prefetch depth PF: columns c+3 .. c+1+PF are in flight / in registers ahead of their use
  Col ahead[PF > 1 ? PF - 1 : 1] = {};
  if (kRing) {
    // lanes outside the image never receive LDS-DMA data: their slices of both slots read as zeros, like the register version's
    typedef typename VecOf<T>::native V4;
    V4* rz = reinterpret_cast<V4*>(ring_mem);
    const V4 zero = {};
#pragma unroll
    for (int k = 0; k < 8; k++) rz[k * kWave + lane] = zero;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (active && xa - 2 >= 0) ldv_o<T, VEC, RAG>(y, off_of(xa - 2), in2.y1, nvalid);     // becomes in1.y1 at the top of the first step
    if (has_col(xa - 1)) ring_issue(xa - 1);
    ring_issue(xa);
  } else if (active) {
    if (xa - 2 >= 0) ldv_o<T, VEC, RAG>(y, off_of(xa - 2), in1.y1, nvalid);
    if (xa - 1 >= 0) load_col(xa - 1, in2);
#pragma unroll
    for (int k = 0; k < PF - 1; k++) if (xa + k < nx && xa + k <= xb + 1) load_col(xa + k, ahead[k]);
  }
  // every lane active and no lane on the first / last image row: the whole strip is interior
  const bool strip_inner = (idx_t)strip * kRowsPerWave - VEC >= 1 && (idx_t)strip * kRowsPerWave + (idx_t)(kWave - 1) * VEC < ny - 1;
  auto step = [&](auto inner, idx_t c) {
    Col pre = {};
    constexpr idx_t kAhead = 2 + PF;
    const bool has_pre = !kRing && c + kAhead < nx && c + kAhead <= xb + 1;
    if (active && has_pre) load_col(c + kAhead, pre);
    const idx_t ca = c + 2, cb = c + 1;
    if (kRing) {
      in1 = in2;
      if (has_col(ca)) ring_fetch(ca, has_col(ca + 1), in2); else in2 = Col{};
      if (has_col(ca + 2)) ring_issue(ca + 2);                 // into the slot just read (the reads above have completed)
    }
    if (ca >= 0 && ca < nx) {                                                                     // stage A
      // lane 0 gets no row above: its first row is never needed (the top halo is its LAST row)
      const T up = lane_up(in2.y2[VEC - 1]);
      primal(inner, ca, in2.y1, in2.y2, up, in1.y1, in2.x, in2.gc, p1, x1_2, kt_2);
    }
    if (cb >= 0 && cb >= xa - 1 && cb < nx) dual(inner, cb, x1_1, x1_2, in1.x, in2.x, in1.y1, in1.y2, p1, y1a_1, y1b_1, false);   // stage B
    if (cb >= xa && cb < nx) {                                                                    // stage C
      const T up = lane_up(y1b_1[VEC - 1]);                      // lane 0: no source, its first row is halo
      primal(inner, cb, y1a_1, y1b_1, up, y1a_0, x1_1, in1.gc, p2, x2_1, kt_c);
      if (owner && cb < xb) {
        stv_o<T, VEC, true, RAG>(x_out, off_of(cb), x2_1, nvalid);
        if (kMid) stv_o<T, VEC, true, RAG>(x_mid, off_of(cb), x1_1, nvalid);
        if (kRes && cb >= rx0 && cb < rx1) {   // dual_residual_transform (backend_pdhg.cu:73-94)
          constexpr bool kEdges = VART && !decltype(inner)::value;
          T sTv[kEdges ? VEC : 1], iTv[kEdges ? VEC : 1];             // sqrt(Tau_j), 1 / (tau sqrt(Tau_j)) of this column's pixels
          if (kEdges) {
#pragma unroll
            for (int j = 0; j < VEC; j++) {
              const idx_t row = row0 + j;
              const int cnt = 4 - (cb == 0 ? 1 : 0) - (cb == nx - 1 ? 1 : 0) - (row == 0 ? 1 : 0) - (row == ny - 1 ? 1 : 0);
              const T sT = cnt == 4 ? sqT : t_sqrt(cnt == 3 ? a.Tcls[1] : a.Tcls[0]);
              sTv[kEdges ? j : 0] = sT; iTv[kEdges ? j : 0] = cnt == 4 ? inv_tauT : (T)1 / (p2.tau * sT);
            }
          }
          if (FAST) {                             // tolerance-compared sums: plain fp32 with fused multiply-adds (see dual)
            T sdd = 0, sdv = 0;
#pragma unroll
            for (int j = 0; j < VEC; j++) {
              const T sT = kEdges ? sTv[kEdges ? j : 0] : sqT, iT = kEdges ? iTv[kEdges ? j : 0] : inv_tauT;
              const T w_hat = t_fma(-sT, kt_1[j], (x1_1[j] - x2_1[j]) * iT);
              const T diff = t_fma(sT, kt_c[j], w_hat);
              if (j < nvalid) { sdd = t_fma(diff, diff, sdd); sdv = t_fma(w_hat, w_hat, sdv); }
            }
            r_dd += (double)sdd; r_dv += (double)sdv;
          } else {
#pragma unroll
            for (int j = 0; j < VEC; j++) {
              const T sT = kEdges ? sTv[kEdges ? j : 0] : sqT;
              const T w_hat = (kEdges && sT != sqT) ? (x1_1[j] - x2_1[j]) / (p2.tau * sT) - sT * kt_1[j] : div_tauT.div(x1_1[j] - x2_1[j]) - sqT * kt_1[j];
              const T diff = w_hat + sT * kt_c[j];
              if (j < nvalid) { r_dd += (double)(diff * diff); r_dv += (double)(w_hat * w_hat); }
            }
          }
        }
      }
    }
    if (c >= xa) {                                                                                // stage D
      T o1[VEC], o2[VEC];
      dual(inner, c, x2_0, x2_1, x1_0, x1_1, y1a_0, y1b_0, p2, o1, o2, true);
      if (owner) {
        stv_o<T, VEC, true, RAG>(y_out, off_of(c), o1, nvalid); stv_o<T, VEC, true, RAG>(y2out, off_of(c), o2, nvalid);
        if (kMid) { stv_o<T, VEC, true, RAG>(y_mid, off_of(c), y1a_0, nvalid); stv_o<T, VEC, true, RAG>(y2mid, off_of(c), y1b_0, nvalid); }
      }
    }
    // shift the pipeline by one column
    if (!kRing) in1 = in2;
    if (kRing) {
    } else if (PF > 1) {
      in2 = ahead[0];
#pragma unroll
      for (int k = 0; k + 1 < PF - 1; k++) ahead[k] = ahead[k + 1];
      ahead[PF > 1 ? PF - 2 : 0] = pre;
    } else in2 = pre;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      x1_0[j] = x1_1[j]; x1_1[j] = x1_2[j];
      y1a_0[j] = y1a_1[j]; y1b_0[j] = y1b_1[j];
      x2_0[j] = x2_1[j];
      kt_1[j] = kt_2[j];
    }
  };
  for (idx_t c = xa - 3; c < xb; c++) {
    // the stencils of this step touch columns c-1 .. c+3 (stage D reads column c+1, stage A column
    // c+1 .. c+2 and their left neighbours): all strictly inside, all four stages running
    if (strip_inner && c >= xa && c >= 2 && c + 3 < nx - 1) step(std::true_type(), c);
    else step(std::false_type(), c);
  }
  if (kRes) {
    r_pd = wave_sum(r_pd); r_pv = wave_sum(r_pv); r_dd = wave_sum(r_dd); r_dv = wave_sum(r_dv);
    if (lane == 0) {
      double* p = partial + 4 * (size_t)blockIdx.x;
      p[0] = r_pd; p[1] = r_pv; p[2] = r_dd; p[3] = r_dv;
    }
  }
}

static bool iter2_desc_ok(const prost_hip_fused_desc* d, int dtype) {
  if (!d || d->is3d || d->L != 1) return false;
  if (d->f_moreau && !(d->f_fn == PROST_FN_ABS && d->g_fn == PROST_FN_SQUARE && d->g_coeff_ptr[1] && !d->g_b_masked)) return false;   // Moreau-wrapped prox_f*: one straight-line instance
  if (d->nx < 4 || d->ny < 4) return false;
  if (d->g_fn < 0 || d->g_fn >= PROST_FN_COUNT || d->f_fn < 0 || d->f_fn >= PROST_FN_COUNT) return false;
  if ((double)d->nx * (double)d->ny * (dtype == 0 ? 4 : 8) >= 4294967296.0) return false;   // 32-bit byte offsets per plane
  // one partial per wavefront of a residual launch must fit the reduction workspace even with one chunk per strip
  if ((d->ny + 62 * (dtype == 0 ? 4 : 2) - 1) / (62 * (dtype == 0 ? 4 : 2)) > (size_t)kReduceBlocks / 2) return false;
  for (int k = 0; k < 7; k++) {
    if (d->f_coeff_ptr[k]) return false;
    if (!aligned16(d->g_coeff_ptr[k])) return false;
  }
  return true;
}

// The straight-line instances exist for the ROF / TV-L1 shapes only: ElemOperation1D<Function1DSquare | Function1DAbs> with scalar
// a = 1, c != 0, d = 0, e = 0 (b scalar or per pixel) and ElemOperationNorm2<Function1DIndLeq0> with scalar
// a = 1, d = 0, e = 0.  Every other combination runs through the run-time dispatched instance, whose
// 4-stage pipeline needs > 240 VGPRs: correct (tests) but ~5x slower than two single launches
// (measured abs / ind_leq0 at 4096^2: 0.79 vs 0.16 ms per iteration), so callers should not pair there.
static bool iter2_fast_shape(const prost_hip_fused_desc* d) {
  if ((d->g_fn != PROST_FN_SQUARE && d->g_fn != PROST_FN_ABS) || d->f_fn != (d->f_moreau ? PROST_FN_ABS : PROST_FN_IND_LEQ0)) return false;
  if (d->g_b_masked && (!d->g_coeff_ptr[1] || d->g_fn != PROST_FN_SQUARE)) return false;     // the merged stream IS the per-pixel b; square data term
  if (d->var_T && (d->g_fn != PROST_FN_SQUARE || !d->g_coeff_ptr[1] || d->g_b_masked)) return false;   // position-dependent Tau: one instance (ROF, per-pixel b)
  for (int k = 0; k < 7; k++) if (d->g_coeff_ptr[k] && k != 1) return false;
  return d->g_coeff_val[0] == 1.0 && d->g_coeff_val[2] != 0.0 && d->g_coeff_val[3] == 0.0 && d->g_coeff_val[4] == 0.0 &&
         d->f_coeff_val[0] == 1.0 && d->f_coeff_val[3] == 0.0 && d->f_coeff_val[4] == 0.0;
}

// The tolerance-class instances (kernel template parameter FMAD) exist for fp32, uniform Tau, prox_f* = norm2:ind_leq0 with a radius
// b > 0 written directly (no Moreau wrap).  A description that asks for them and is not of that shape runs the exact instances: exact
// results satisfy every tolerance.
static bool iter2_fmad_shape(const prost_hip_fused_desc* d, int dtype) {
  return d->arith == PROST_HIP_ARITH_FMAD && dtype == 0 && iter2_fast_shape(d) && !d->var_T && !d->f_moreau && d->f_fn == PROST_FN_IND_LEQ0 &&
         d->f_coeff_val[1] > 0.0;
}

// chunk length (columns per wavefront) of a launch; `res`: the launch also forms the residual sums
static int iter2_chunk_cols(const prost_hip_fused_desc* d, int V, bool res, int cols) {
  const size_t strips = (d->ny + 62 * V - 1) / (62 * V);
  // resident wavefronts per SIMD of the instance that will run: LDS-ring instances 4 (3 with the residual sums), register-ring
  // instances 3 (2)
  static const bool no_ring = []() { const char* e = getenv("PROST_ITER2_NO_RING"); return e && atoi(e) != 0; }();
  const bool ring = !no_ring && iter2_fast_shape(d) && d->ny % (size_t)V == 0;
  if (cols <= 0) {
    // The kernel is bound by wave-level latency as much as by HBM.  256 CUs x 4 SIMDs x resident waves: take the longest
    // chunk (3 warm-up columns are amortised over it) that still fills >= 90 % of the slots in ONE round -- a second,
    // mostly empty round costs a full chunk time (register ring, 3 waves / SIMD: 24 cols = 2907 waves 0.117 ms, 18 cols =
    // 3876 waves 0.127 ms; LDS ring, 4 waves / SIMD: 18 and 24 cols 0.100 ms, 36 cols 0.117; with the residual sums, 3
    // waves: 24 cols 0.116 ms, 18 cols 0.121, 36 cols 0.122).  Chunk lengths stay off multiples of 16 (HBM channel spread).
    const size_t slots = 256 * 4 * (size_t)(ring ? (res ? 3 : 4) : (res ? 2 : 3));
    cols = 0;
    for (int c : {36, 30, 24, 18, 12, 9}) if (strips * ((d->nx + c - 1) / c) * 10 >= slots * 9) { cols = c; break; }
    if (cols == 0) {
      // small images cannot fill the chip: the launch then lasts as long as ONE wave needs for its chunk
      // (c + 3 pipeline steps), so the shortest chunk that still fits one round wins (256^2: 1 column
      // 0.009 ms, 6 columns 0.019 ms; 1024^2: 2-3 columns)
      double best = 1e30;
      for (int c : {18, 12, 9, 6, 4, 3, 2, 1}) {
        const double waves = (double)(strips * ((d->nx + c - 1) / c));
        const double cost = (c + 3.5) * (waves > (double)slots ? waves / (double)slots : 1.0);
        if (cost < best) { best = cost; cols = c; }
      }
    }
  }
  // residual launches write one partial (4 doubles) per wavefront: kReduceBlocks / 2 of them fit the workspace
  while (res && (size_t)cols < d->nx && strips * ((d->nx + cols - 1) / cols) > (size_t)kReduceBlocks / 2) cols += 6;
  return cols;
}

template <class T>
static int run_iter2(const prost_hip_fused_desc* d, T* x_out, T* y_out, const T* x, const T* y, T* x_mid, T* y_mid, const double* tau,
                     const double* sigma, const double* theta, int cols, double* out4, void* ws, void* stream, void* record = nullptr, const RuleTail* tail = nullptr) {
  constexpr int V = VecOf<T>::N;
  if (!iter2_desc_ok(d, sizeof(T) == 4 ? 0 : 1) || !aligned16(x_out) || !aligned16(y_out) || !aligned16(x) || !aligned16(y)) {
    set_error("fused double iteration: unsupported description"); return 1;
  }
  if ((x_mid == nullptr) != (y_mid == nullptr) || !aligned16(x_mid) || !aligned16(y_mid)) { set_error("fused double iteration: x_mid and y_mid go together"); return 1; }
  if (out4 && !ws) { set_error("fused double iteration: residuals need the reduction workspace"); return 1; }
  FusedArgs<T> a = make_fused_args<T>(d);
  const size_t strips = (d->ny + 62 * V - 1) / (62 * V);
  cols = iter2_chunk_cols(d, V, out4 != nullptr, cols);
  if (out4 && strips * ((d->nx + cols - 1) / cols) > (size_t)kReduceBlocks / 2) { set_error("fused double iteration: grid exceeds the reduction workspace"); return 1; }
  a.cols_per_block = cols;
  a.chunks = (unsigned)((d->nx + cols - 1) / cols);
  if (strips * a.chunks > 0x7fffffffull) { set_error("fused double iteration: grid too large"); return 1; }
  const PdhgRecord<T>* rec = static_cast<const PdhgRecord<T>*>(record);
  // with a device record the step sizes are not known here: the dispatch below may only depend on the coefficients.  e = 0 on
  // both sides makes the fp64 denominators exactly 1 for EVERY step size, which is what the straight-line instances assume
  const double one2[2] = {1.0, 1.0};
  if (rec) { tau = sigma = theta = one2; }
  if (rec && (d->g_coeff_val[4] != 0.0 || d->f_coeff_val[4] != 0.0 || d->g_coeff_ptr[4])) { set_error("fused double iteration: device-resident step sizes need e = 0"); return 1; }
  IterParams<T> p[2];
  for (int i = 0; i < 2; i++) {
    p[i].tau = (T)tau[i]; p[i].sigma = (T)sigma[i]; p[i].theta = (T)theta[i];
    p[i].ug = make_uniform_prox<T>(a.g_val, (T)tau[i] * a.Tval);
    p[i].uf = make_uniform_prox<T>(a.f_val, dual_prox_step<T>((T)sigma[i], a.Sval, a.fmor));
    for (int k = 0; k < 2; k++) p[i].ec[k] = a.varT ? make_edge_terms<T>(a.g_val, (T)tau[i] * a.Tcls[k]) : EdgeTerms<T>();
  }
  int mask = 0;
  for (int k = 0; k < 7; k++) if (d->g_coeff_ptr[k]) mask |= 1 << k;
  // straight-line instance for the ROF shape: square / ind_leq0 with scalar a = 1, d = 0, e = 0 on both
  // sides (so a (v - d tau) = v and the fp64 denominators are exactly 1), b of prox_g per pixel
  if (d->g_b_masked && !iter2_fast_shape(d)) { set_error("fused double iteration: a merged b stream needs the straight-line square shape"); return 1; }
  const bool fast = iter2_fast_shape(d) && p[0].ug.a_one && p[0].ug.den_one && !p[0].ug.degenerate && p[0].uf.a_one && p[0].uf.den_one &&
                    p[1].ug.den_one && p[1].uf.den_one;
  if (a.fmor && !(fast && mask == 0x2)) { set_error("fused double iteration: a Moreau-wrapped prox_f* runs the straight-line square / abs instance only"); return 1; }
  const bool fmad = fast && iter2_fmad_shape(d, sizeof(T) == 4 ? 0 : 1);
  dim3 grid((unsigned)(strips * a.chunks)), block(kWave);
  hipStream_t s = as_stream(stream);
  const int mode = (out4 ? 2 : 0) | (x_mid ? 1 : 0);
  const bool rag = d->ny % V != 0;
  double* partial = static_cast<double*>(ws);
#define GO5(G, F, M, PFv, FASTv, MODEv, RAGv, VARTv, FMADv) \
  PH_LAUNCH((fused_iter2d_x2_kernel<T, V, G, F, M, PFv, FASTv, MODEv, RAGv, VARTv, FMADv>), grid, block, 0, s, x_out, y_out, x, y, x_mid, y_mid, a, p[0], p[1], partial, rec)
// position-dependent Tau (FusedArgs::varT): instances exist for the straight-line ROF shape with a per-pixel b (square, mask 0x2) -- the
// shape callers pair on (prost_hip_fused_iteration2_profitable); the others are refused
// tolerance-class arithmetic (`fmad`, decided above): fp32 straight-line instances with prox_f* = norm2:ind_leq0
#define GO4(G, F, M, PFv, FASTv, MODEv, RAGv) do { \
    if (a.varT) { \
      if constexpr (FASTv && G == PROST_FN_SQUARE && M == 0x2) GO5(G, F, M, PFv, FASTv, MODEv, RAGv, true, false); \
      else { set_error("fused double iteration: no position-dependent Tau instance for this shape"); return 1; } \
    } else if (fmad) { \
      if constexpr (FASTv && F == PROST_FN_IND_LEQ0 && sizeof(T) == 4) GO5(G, F, M, PFv, FASTv, MODEv, RAGv, false, true); \
      else { set_error("fused double iteration: no tolerance-class instance for this shape"); return 1; } \
    } else GO5(G, F, M, PFv, FASTv, MODEv, RAGv, false, false); \
  } while (0)
// straight-line instances of heights that are a multiple of the vector width prefetch through the LDS ring (PF = 0); ragged
// heights (4-byte aligned column starts) keep the register ring
// (PROST_ITER2_NO_RING=1 forces the register ring everywhere: A/B measurements)
  static const bool no_ring = []() { const char* e = getenv("PROST_ITER2_NO_RING"); return e && atoi(e) != 0; }();
#define GO3(G, F, M, PFv, FASTv, MODEv) do { if (rag) GO4(G, F, M, PFv, FASTv, MODEv, true); else if (no_ring) GO4(G, F, M, PFv, FASTv, MODEv, false); \
    else GO4(G, F, M, (FASTv ? 0 : PFv), FASTv, MODEv, false); } while (0)
#define GO(G, F, M, PFv, FASTv) do { if (mode == 0) GO3(G, F, M, PFv, FASTv, 0); else if (mode == 1) GO3(G, F, M, PFv, FASTv, 1); \
    else if (mode == 2) GO3(G, F, M, PFv, FASTv, 2); else GO3(G, F, M, PFv, FASTv, 3); } while (0)
  if (fast && d->g_fn == PROST_FN_ABS) { if (mask == 0x2) GO(PROST_FN_ABS, PROST_FN_IND_LEQ0, 0x2, 3, true); else GO(PROST_FN_ABS, PROST_FN_IND_LEQ0, 0, 3, true); }
  else if (fast && mask == 0x2 && d->g_b_masked) GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, 0x82, 3, true);      // inpainting: binary mask folded into b
  else if (fast && mask == 0x2 && a.fmor) GO(PROST_FN_SQUARE, PROST_FN_ABS, 0x2, 3, true);          // ROF in its primal form (Moreau-wrapped TV norm)
  else if (fast && mask == 0x2) {
    GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, 0x2, 3, true);
  }
  else if (fast) GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, 0, 3, true);                 // b of prox_g is a scalar too
  else if (d->g_fn == PROST_FN_SQUARE && d->f_fn == PROST_FN_IND_LEQ0 && mask == 0x2) GO(PROST_FN_SQUARE, PROST_FN_IND_LEQ0, 0x2, 1, false);
  else if (mask == 0) GO(-1, -1, 0, 1, false);
  else GO(-1, -1, 0x7F, 1, false);
#undef GO
#undef GO3
#undef GO4
#undef GO5
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, "fused double iteration kernel"); }
  if (out4 && tail && tail->apply) return launch_fold4_rule<T>(out4, partial, grid.x, record, tail->iteration, tail->mirror, s);
  if (out4) return launch_fold4(out4, partial, grid.x, s);
  return 0;
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {
int prost_hip_fused_iteration2_supported(const prost_hip_fused_desc* desc, int dtype) { return iter2_desc_ok(desc, dtype) ? 1 : 0; }
int prost_hip_fused_iteration2_profitable(const prost_hip_fused_desc* desc, int dtype) { return iter2_desc_ok(desc, dtype) && iter2_fast_shape(desc) ? 1 : 0; }
int prost_hip_fused_iteration2_arith(const prost_hip_fused_desc* desc, int dtype) {
  return iter2_desc_ok(desc, dtype) && iter2_fmad_shape(desc, dtype) ? PROST_HIP_ARITH_FMAD : PROST_HIP_ARITH_EXACT;
}
int prost_hip_fused_iteration2_chunk_cols(const prost_hip_fused_desc* desc, int dtype, int with_residuals) {
  return iter2_desc_ok(desc, dtype) ? iter2_chunk_cols(desc, dtype == 0 ? 4 : 2, with_residuals != 0, 0) : 0;
}
int prost_hip_fused_iteration2_f32(const prost_hip_fused_desc* d, float* x_out, float* y_out, const float* x, const float* y, float* x_mid, float* y_mid,
                                   const double* tau, const double* sigma, const double* theta, int cols_per_block, double* res_out4, void* workspace, void* s) {
  return run_iter2<float>(d, x_out, y_out, x, y, x_mid, y_mid, tau, sigma, theta, cols_per_block, res_out4, workspace, s);
}
int prost_hip_fused_iteration2_f64(const prost_hip_fused_desc* d, double* x_out, double* y_out, const double* x, const double* y, double* x_mid, double* y_mid,
                                   const double* tau, const double* sigma, const double* theta, int cols_per_block, double* res_out4, void* workspace, void* s) {
  return run_iter2<double>(d, x_out, y_out, x, y, x_mid, y_mid, tau, sigma, theta, cols_per_block, res_out4, workspace, s);
}
int prost_hip_fused_iteration2_rec_f32(const prost_hip_fused_desc* d, float* x_out, float* y_out, const float* x, const float* y, float* x_mid, float* y_mid,
                                       void* record, int cols_per_block, double* res_out4, void* workspace, int apply_rule, unsigned long long iteration,
                                       prost_hip_pdhg_rule_state* mirror, void* s) {
  if (!record) { set_error("fused double iteration: no step-size record"); return 1; }
  const RuleTail tail = {apply_rule, iteration, mirror};
  return run_iter2<float>(d, x_out, y_out, x, y, x_mid, y_mid, nullptr, nullptr, nullptr, cols_per_block, res_out4, workspace, s, record, &tail);
}
int prost_hip_fused_iteration2_rec_f64(const prost_hip_fused_desc* d, double* x_out, double* y_out, const double* x, const double* y, double* x_mid, double* y_mid,
                                       void* record, int cols_per_block, double* res_out4, void* workspace, int apply_rule, unsigned long long iteration,
                                       prost_hip_pdhg_rule_state* mirror, void* s) {
  if (!record) { set_error("fused double iteration: no step-size record"); return 1; }
  const RuleTail tail = {apply_rule, iteration, mirror};
  return run_iter2<double>(d, x_out, y_out, x, y, x_mid, y_mid, nullptr, nullptr, nullptr, cols_per_block, res_out4, workspace, s, record, &tail);
}
}  // extern "C"
