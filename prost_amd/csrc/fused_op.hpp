// fused_op.hpp -- a linear operator made of sparse and gradient blocks, applied INSIDE other kernels (round 3: the stage kernels of the
// ADMM graph projection, kernels_cgls.hip; round 5: the prox kernels of the generic PDHG iteration, kernels_prox.hip): the thread that
// owns VEC consecutive output elements evaluates their rows (K) or columns (K^T) for every block that covers them, in block order, with
// the expressions and the summation order of the stand-alone products (csr_spmv_kernel<., 1, .> / pattern_spmv_kernel / grad_fwd_kernel /
// grad_adj_kernel: LinearOperator::Eval, linearoperator.cu:135-170) -- the product never reaches memory.
#pragma once
#include "fused_common.hpp"

namespace prost_hip {

struct OpBlockDev {
  int kind;
  unsigned long long row, col, nrows, ncols, nx, ny, L;
  const void* val; const int32_t* ptr; const int32_t* ind;
  const void* val_t; const int32_t* ptr_t; const int32_t* ind_t;
  // sparse blocks whose product runs from row patterns (prost_hip_pattern_spmv): one 16-bit pattern number per row + the table
  // (entries pptr[id] .. pptr[id + 1] - 1: column - row in rel, value in pval); ids / ids_t null: the CSR arrays above
  const uint16_t* ids; const int32_t* pptr; const int32_t* rel; const void* pval;
  const uint16_t* ids_t; const int32_t* pptr_t; const int32_t* rel_t; const void* pval_t;
  const int32_t* anchor; const int32_t* anchor_t;      // anchored tables (pattern_rows); null: offsets count from the row number
};
struct FusedOpDev { int nblocks; OpBlockDev b[PROST_HIP_OP_MAX_BLOCKS]; };

// The block table and the pattern tables are written by the host before the first launch and never by a kernel.  Kernels read them
// through the CONSTANT address space: a wave-uniform load from there is a scalar load (batched, results in SGPRs) whatever the kernel has
// stored before.  Through a generic pointer the compiler must assume that the kernel's own stores may alias the table (the grid-stride
// loop stores results before it walks the table again) and emits per-lane VECTOR loads, each field waited for before the next is
// requested: 8-10 dependent memory round trips per block and wavefront (measured: 100-120 us for a prox launch at 2048^2 that takes 35 with
// scalar loads).
#define PROST_CONSTANT __attribute__((address_space(4)))
template <class U> __device__ __forceinline__ const PROST_CONSTANT U* as_constant(const U* p) { return (const PROST_CONSTANT U*)p; }
template <class U> __device__ __forceinline__ const PROST_CONSTANT U* as_constant_of(const void* p) { return (const PROST_CONSTANT U*)p; }

// a[j] belongs to element j * 64 + lane of a 64 VEC-element range; out[c] := element VEC * lane + c (all 64 lanes active)
template <class T, int VEC>
__device__ __forceinline__ void wave_untranspose(const T (&a)[VEC], T (&out)[VEC], unsigned lane) {
  const unsigned src_j = (VEC * lane) / kWave;          // the same for the VEC elements of a lane: VEC divides 64
#pragma unroll
  for (int c = 0; c < VEC; c++) {
    const int src_lane = (int)((VEC * lane + c) & (kWave - 1));
    T v = 0;
#pragma unroll
    for (int j = 0; j < VEC; j++) { const T tmp = __shfl(a[j], src_lane, kWave); if ((unsigned)j == src_j) v = tmp; }
    out[c] = v;
  }
}

// sum[j] = (A rhs)_(r_j), j < VEC, for VEC rows of a CSR matrix -- ONE loop over the entry positions with the VEC rows side
// by side, so that the loads of the VEC rows (row starts, then values / indices, then gathered operands: three dependent
// levels) are in flight together instead of one row after the other.  Per row the entries are summed in order, as
// csr_spmv_kernel<T, 1, .> does.
template <class T, int VEC>
__device__ __forceinline__ void csr_rows(const T* __restrict__ val, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ind,
                                         const T* __restrict__ rhs, const size_t (&r)[VEC], T (&sum)[VEC]) {
  int32_t b[VEC], e[VEC], len = 0;
#pragma unroll
  for (int j = 0; j < VEC; j++) { b[j] = ptr[r[j]]; e[j] = ptr[r[j] + 1]; }
#pragma unroll
  for (int j = 0; j < VEC; j++) { sum[j] = 0; len = e[j] - b[j] > len ? e[j] - b[j] : len; }
  for (int32_t st = 0; st < len; st++) {
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const int32_t k = b[j] + st;
      if (k < e[j]) sum[j] += val[k] * rhs[ind[k]];
    }
  }
}
// the contribution of one CSR block (or its transpose) to the VEC elements starting at local index r0 of this lane; `wave0`: local
// index of lane 0's first element, `whole`: the wavefront's 64 VEC elements all lie inside the block -- then the lanes take the rows
// transposed (lane, lane + 64, ...: neighbouring lanes load neighbouring rows) and the sums are shuffled back
template <class T, int VEC>
__device__ __forceinline__ void csr_contrib(const T* __restrict__ val, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ind,
                                            const T* __restrict__ rhs, size_t r0, size_t wave0, bool whole, T (&sum)[VEC]) {
  size_t r[VEC];
  if (VEC > 1 && whole) {
    const unsigned lane = threadIdx.x & (kWave - 1);
    T st[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) r[j] = wave0 + (size_t)j * kWave + lane;
    csr_rows<T, VEC>(val, ptr, ind, rhs, r, st);
    wave_untranspose<T, VEC>(st, sum, lane);
  } else {
#pragma unroll
    for (int j = 0; j < VEC; j++) r[j] = r0 + j;
    csr_rows<T, VEC>(val, ptr, ind, rhs, r, sum);
  }
}

// sum[j] = (A rhs)_(r0 + j) for VEC consecutive rows of a pattern-compressed matrix: the entries of a row's pattern in order, as
// pattern_spmv_kernel (out = 0 ; out += value * rhs[row + rel]) -- the bits of the CSR product.  Nearly always the VEC rows of a lane, and
// the rows of the whole wavefront, have ONE pattern: its table entries are then wave-uniform (scalar loads) and an entry costs one load
// of VEC consecutive operands per lane (element-aligned 16-byte accesses, which gfx950 serves); a wavefront on the seams of a
// stencil takes one pass per pattern it holds.
template <class T, int VEC> struct OpPack { typedef T V __attribute__((ext_vector_type(VEC), aligned(sizeof(T)))); };
// (IP / VP: where the table lives -- the constant address space: scalar loads; or LDS, staged by the workgroup: broadcast reads)
template <class T, int VEC, int NR, class IP, class VP>
__device__ __forceinline__ void pattern_rows(const uint16_t* __restrict__ ids, IP pptr, IP rel, VP pval, const T* const (&rhs)[NR], size_t r0, T (&sum)[NR][VEC],
                                             const int32_t* __restrict__ anchor = nullptr) {
  typedef typename OpPack<T, VEC>::V PV;
  constexpr int kB = NR == 1 ? 6 : 4;                              // entries per batch: kB * NR operand loads of VEC elements in flight
  unsigned id[VEC];
  if (VEC == 4) {                                                  // r0 is a multiple of 4 and the numbers start on an 8-byte boundary
    const uint2 w = *reinterpret_cast<const uint2*>(ids + r0);
    id[0] = w.x & 0xFFFFu; id[1 % VEC] = w.x >> 16; id[2 % VEC] = w.y & 0xFFFFu; id[3 % VEC] = w.y >> 16;
  } else {
#pragma unroll
    for (int j = 0; j < VEC; j++) id[j] = ids[r0 + j];
  }
  // ANCHORED tables (round 5; anchor != null): the offsets of a pattern count from anchor[row] -- the row's first column -- instead of from
  // the row number, so matrices whose range and domain have different geometry (convmtx2's FULL convolution of example_deblurring.m: the
  // output image is larger than the input, column - row drifts by 14 per image column) repeat their patterns as well.  Consecutive rows of
  // an image column have consecutive anchors: the 16-byte operand loads stay, a lane whose anchors are not consecutive goes row by row.
  long base[VEC];
  bool consec = true;
  if (anchor) {
    int32_t an[VEC];
    if (VEC == 4) { const int4 w = *reinterpret_cast<const int4*>(anchor + r0); an[0] = w.x; an[1 % VEC] = w.y; an[2 % VEC] = w.z; an[3 % VEC] = w.w; }
    else {
#pragma unroll
      for (int j = 0; j < VEC; j++) an[j] = anchor[r0 + j];
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) { base[j] = (long)an[j]; consec = consec && an[j] == an[0] + j; }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; j++) base[j] = (long)(r0 + j);
  }
#pragma unroll
  for (int q = 0; q < NR; q++)
#pragma unroll
    for (int j = 0; j < VEC; j++) sum[q][j] = 0;
  if (VEC > 1) {
    // One pass per DISTINCT pattern among the wavefront's rows (one in the interior of a stencil, two or three on its seams): the pattern
    // number of a pass is wave-uniform, so its table entries are
    // scalar loads and the operands of an entry are requested by all lanes together -- a lane whose VEC rows all have the pattern with
    // one 16-byte load, a lane on the seam row by row.  Every row still sums the entries of ITS pattern in table order.
    unsigned todo = (1u << VEC) - 1;                               // rows of this lane whose pattern has not had its pass
    for (;;) {
      unsigned cand = 0;
#pragma unroll
      for (int j = VEC - 1; j >= 0; j--) cand = (todo >> j & 1) ? id[j] : cand;
      const unsigned long long open = __builtin_amdgcn_ballot_w64(todo != 0);
      if (open == 0) break;
      const unsigned cur = (unsigned)__builtin_amdgcn_readlane((int)cand, (int)__builtin_ctzll(open));
      bool mine[VEC], all = true;
      unsigned mask = 0;
#pragma unroll
      for (int j = 0; j < VEC; j++) { mine[j] = id[j] == cur; all = all && mine[j]; mask |= mine[j] ? 1u << j : 0u; }
      todo &= ~mask;
      const int32_t b = pptr[cur], e = pptr[cur + 1];
      for (int32_t k0 = b; k0 < e; k0 += kB) {
        long r[kB]; T v[kB];
#pragma unroll
        for (int u = 0; u < kB; u++) { const int32_t k = k0 + u < e ? k0 + u : e - 1; r[u] = (long)rel[k]; v[u] = pval[k]; }
        PV x[NR][kB];
        if (all && consec) {
#pragma unroll
          for (int u = 0; u < kB; u++)
#pragma unroll
            for (int q = 0; q < NR; q++) x[q][u] = *reinterpret_cast<const PV*>(rhs[q] + base[0] + r[u]);
        } else {
#pragma unroll
          for (int u = 0; u < kB; u++)
#pragma unroll
            for (int q = 0; q < NR; q++)
#pragma unroll
              for (int j = 0; j < VEC; j++) x[q][u][j] = rhs[q][mine[j] ? base[j] + r[u] : 0L];     // (element 0: always there, never used)
        }
#pragma unroll
        for (int u = 0; u < kB; u++) {
          if (k0 + u < e) {
#pragma unroll
            for (int q = 0; q < NR; q++)
#pragma unroll
              for (int j = 0; j < VEC; j++) sum[q][j] = mine[j] ? sum[q][j] + v[u] * x[q][u][j] : sum[q][j];
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    const int32_t b = pptr[id[j]], e = pptr[id[j] + 1];
#pragma unroll
    for (int q = 0; q < NR; q++) {
      T s = 0;
      for (int32_t k = b; k < e; k++) s += pval[k] * rhs[q][base[j] + (long)rel[k]];
      sum[q][j] = s;
    }
  }
}

// (lanes_in_step = false: the caller's lanes do not take the 64 VEC consecutive elements of a wavefront step together -- the rows of a
// CSR block are then taken lane by lane, never transposed across the wavefront)
// kv[0..VEC) = (K rhs)_(i .. i+VEC).  VEC > 1: i, every block's row / col / nrows and every gradient block's ny and plane
// size are multiples of VEC (host-checked), so the VEC rows lie in the same blocks, the same component plane and image column.
// w0 = the i of lane 0 (wave-uniform); VEC > 1 callers guarantee that all 64 lanes are active.
// The header of a block as ONE batch of scalar loads: the fields are requested together and pinned in SGPRs before the first branch looks
// at one of them.  Left to itself the compiler sinks every load to its first use behind the branch in front of it -- load, wait, compare,
// branch, load, wait ... six or seven scalar round trips per block and wavefront before the first operand is requested (the prox launches
// of the generic path spent two thirds of a wavefront's life in s_waitcnt lgkmcnt(0)).
struct OpHead { int kind; unsigned long long row, col, nrows, ncols, nx, ny, L; const uint16_t* ids; const int32_t* pptr; const int32_t* rel; const void* pval; const int32_t* anchor; };
template <bool ADJ>
__device__ __forceinline__ OpHead op_head(const PROST_CONSTANT OpBlockDev& B) {
  OpHead h;
  h.kind = B.kind; h.row = B.row; h.col = B.col; h.nrows = B.nrows; h.ncols = B.ncols; h.nx = B.nx; h.ny = B.ny; h.L = B.L;
  h.ids = ADJ ? B.ids_t : B.ids; h.pptr = ADJ ? B.pptr_t : B.pptr; h.rel = ADJ ? B.rel_t : B.rel; h.pval = ADJ ? B.pval_t : B.pval;
  h.anchor = ADJ ? B.anchor_t : B.anchor;
  asm volatile("" : : "s"(h.kind), "s"(h.row), "s"(h.col), "s"(h.nrows), "s"(h.ncols), "s"(h.nx), "s"(h.ny), "s"(h.L), "s"(h.ids), "s"(h.pptr), "s"(h.rel),
               "s"(h.pval), "s"(h.anchor));
  return h;
}

// ZERO = false: kv holds the vector the product is ADDED to (LinearOperator::Eval with accumulate: EvalAdd per block, in block order, onto
// what is there -- (v + K_1 t) + K_2 t, not v + (K_1 t + K_2 t): the two differ in the last place where two blocks share rows)
template <class T, int VEC, int NR, bool ZERO = true>
__device__ __forceinline__ void op_fwd_rows_n(const PROST_CONSTANT FusedOpDev& op, size_t i, size_t w0, const T* const (&t)[NR], T (&kv)[NR][VEC], bool lanes_in_step = true) {
  if (ZERO) {
#pragma unroll
    for (int q = 0; q < NR; q++)
#pragma unroll
      for (int j = 0; j < VEC; j++) kv[q][j] = 0;
  }
  for (int b = 0; b < op.nblocks; b++) {
    const PROST_CONSTANT OpBlockDev& B = op.b[b];
    const OpHead H = op_head<false>(B);
    if ((i < H.row) | (i >= H.row + H.nrows)) continue;
    const size_t r = i - H.row;
    if (H.kind == PROST_OP_CSR) {
      const bool whole = VEC > 1 && lanes_in_step && w0 >= H.row && w0 + (size_t)kWave * VEC <= H.row + H.nrows;
      T sum[NR][VEC];
      if (H.ids) {
        const T* rhs[NR];
#pragma unroll
        for (int q = 0; q < NR; q++) rhs[q] = t[q] + H.col;
        pattern_rows<T, VEC, NR>(H.ids, as_constant(H.pptr), as_constant(H.rel), as_constant_of<T>(H.pval), rhs, r, sum, H.anchor);
      } else {
#pragma unroll
        for (int q = 0; q < NR; q++) csr_contrib<T, VEC>(static_cast<const T*>(B.val), B.ptr, B.ind, t[q] + H.col, r, w0 - H.row, whole, sum[q]);
      }
#pragma unroll
      for (int q = 0; q < NR; q++)
#pragma unroll
        for (int j = 0; j < VEC; j++) kv[q][j] = kv[q][j] + sum[q][j];
    } else {
      const unsigned nx = (unsigned)H.nx, ny = (unsigned)H.ny, slice = nx * ny, N = slice * (unsigned)H.L;
      const unsigned r32 = (unsigned)r;
      const unsigned c = r32 / N, idx = r32 - c * N;
#pragma unroll
      for (int q = 0; q < NR; q++) {
        const T* rhs = t[q] + H.col;
        T cur[VEC], g[VEC];
        ldv<T, VEC>(rhs + idx, cur);
        if (c == 0) {
          const unsigned x = (idx / ny) % nx;
          T nb[VEC];
#pragma unroll
          for (int j = 0; j < VEC; j++) nb[j] = 0;
          if (x < nx - 1) ldv<T, VEC>(rhs + idx + ny, nb);
#pragma unroll
          for (int j = 0; j < VEC; j++) g[j] = x < nx - 1 ? nb[j] - cur[j] : (T)0;
        } else if (c == 1) {
          const unsigned y = idx % ny;
          const T below = y + VEC < ny ? rhs[idx + VEC] : (T)0;
#pragma unroll
          for (int j = 0; j < VEC; j++) {
            const T dn = j + 1 < VEC ? cur[(j + 1) % VEC] : below;
            g[j] = y + j < ny - 1 ? dn - cur[j] : (T)0;
          }
        } else {
          const unsigned l = idx / slice;
          T up[VEC];
#pragma unroll
          for (int j = 0; j < VEC; j++) up[j] = 0;
          if (l < (unsigned)H.L - 1) ldv<T, VEC>(rhs + idx + slice, up);
#pragma unroll
          for (int j = 0; j < VEC; j++) g[j] = l < (unsigned)H.L - 1 ? up[j] - cur[j] : -cur[j];      // Dirichlet (block_gradient3d.cu:73-76)
        }
#pragma unroll
        for (int j = 0; j < VEC; j++) kv[q][j] = kv[q][j] + g[j];
      }
    }
  }
}
template <class T, int VEC, bool ZERO = true>
__device__ __forceinline__ void op_fwd_rows(const PROST_CONSTANT FusedOpDev& op, size_t i, size_t w0, const T* __restrict__ t, T (&kv)[VEC], bool lanes_in_step = true) {
  const T* tt[1] = {t};
  T k1[1][VEC];
  if (!ZERO) {
#pragma unroll
    for (int j = 0; j < VEC; j++) k1[0][j] = kv[j];
  }
  op_fwd_rows_n<T, VEC, 1, ZERO>(op, i, w0, tt, k1, lanes_in_step);
#pragma unroll
  for (int j = 0; j < VEC; j++) kv[j] = k1[0][j];
}
// v[0..VEC) += (K^T rhs)_(j .. j+VEC), blocks in order (EvalAdjointAdd per block)
template <class T, int VEC>
__device__ __forceinline__ void op_adj_cols(const PROST_CONSTANT FusedOpDev& op, size_t jg, size_t w0, const T* __restrict__ t, T (&v)[VEC], bool lanes_in_step = true) {
  for (int b = 0; b < op.nblocks; b++) {
    const PROST_CONSTANT OpBlockDev& B = op.b[b];
    const OpHead H = op_head<true>(B);
    if ((jg < H.col) | (jg >= H.col + H.ncols)) continue;
    const size_t cidx = jg - H.col;
    const T* rhs = t + H.row;
    if (H.kind == PROST_OP_CSR) {
      const bool whole = VEC > 1 && lanes_in_step && w0 >= H.col && w0 + (size_t)kWave * VEC <= H.col + H.ncols;
      T sum[1][VEC];
      if (H.ids) {
        const T* rr[1] = {rhs};
        pattern_rows<T, VEC, 1>(H.ids, as_constant(H.pptr), as_constant(H.rel), as_constant_of<T>(H.pval), rr, cidx, sum, H.anchor);
      } else csr_contrib<T, VEC>(static_cast<const T*>(B.val_t), B.ptr_t, B.ind_t, rhs, cidx, w0 - H.col, whole, sum[0]);
#pragma unroll
      for (int j = 0; j < VEC; j++) v[j] = v[j] + sum[0][j];
    } else {
      const unsigned nx = (unsigned)H.nx, ny = (unsigned)H.ny, slice = nx * ny, idx = (unsigned)cidx;
      const size_t N = (size_t)slice * H.L;
      const unsigned y = idx % ny, x = (idx / ny) % nx;
      T px[VEC], pxm[VEC], py[VEC];
      ldv<T, VEC>(rhs + idx, px);
      ldv<T, VEC>(rhs + N + idx, py);
#pragma unroll
      for (int j = 0; j < VEC; j++) pxm[j] = 0;
      if (x > 0) ldv<T, VEC>(rhs + idx - ny, pxm);
      const T above = y > 0 ? rhs[N + idx - 1] : (T)0;
      T pl[VEC], plm[VEC];
      unsigned l = 0;
      if (H.kind == PROST_OP_GRAD3D) {
        l = idx / slice;
        ldv<T, VEC>(rhs + 2 * N + idx, pl);
#pragma unroll
        for (int j = 0; j < VEC; j++) plm[j] = 0;
        if (l > 0) ldv<T, VEC>(rhs + 2 * N + idx - slice, plm);
      }
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        T divx, divy;
        if (y + j < ny - 1) divy = py[j]; else divy = 0;
        if (y + j > 0) divy -= j > 0 ? py[(j + VEC - 1) % VEC] : above;
        if (x < nx - 1) divx = px[j]; else divx = 0;
        if (x > 0) divx -= pxm[j];
        T sdiv;
        if (H.kind == PROST_OP_GRAD3D) {
          T divl = pl[j];
          if (l > 0) divl -= plm[j];
          sdiv = divx + divy + divl;
        } else {
          sdiv = divx + divy;
        }
        v[j] = v[j] - sdiv;                     // adjoint is minus the divergence
      }
    }
  }
}

inline bool fused_op_ok(const prost_hip_fused_op* op, uint64_t m, uint64_t n) {
  if (!op || op->nblocks < 1 || op->nblocks > PROST_HIP_OP_MAX_BLOCKS) return false;
  for (int b = 0; b < op->nblocks; b++) {
    const prost_hip_op_block& B = op->block[b];
    if (B.nrows == 0 || B.ncols == 0 || B.row + B.nrows > m || B.col + B.ncols > n) return false;
    if (B.kind == PROST_OP_CSR) {
      if (!(B.ids ? (B.pptr && B.rel && B.pval) : (B.val && B.ptr && B.ind))) return false;                 // K: row patterns or CSR
      if (!(B.ids_t ? (B.pptr_t && B.rel_t && B.pval_t) : (B.val_t && B.ptr_t && B.ind_t))) return false;   // K^T likewise
    } else if (B.kind == PROST_OP_GRAD2D || B.kind == PROST_OP_GRAD3D) {
      const uint64_t N = B.nx * B.ny * B.L, comps = B.kind == PROST_OP_GRAD2D ? 2 : 3;
      if (B.nx == 0 || B.ny == 0 || B.L == 0 || N != B.ncols || comps * N != B.nrows) return false;
      if (comps * N >= ((uint64_t)1 << 32)) return false;                // 32-bit element offsets inside a block
    } else {
      return false;
    }
  }
  return true;
}
// VEC rows / columns per thread need every block boundary, gradient height and plane size on a multiple of VEC
inline bool fused_op_vec_ok(const prost_hip_fused_op* op, unsigned V) {
  for (int b = 0; b < op->nblocks; b++) {
    const prost_hip_op_block& B = op->block[b];
    if (B.row % V || B.col % V || B.nrows % V || B.ncols % V) return false;
    if (B.kind != PROST_OP_CSR && (B.ny % V || (B.nx * B.ny * B.L) % V)) return false;
  }
  return true;
}
inline FusedOpDev make_op(const prost_hip_fused_op* op) {
  FusedOpDev o;
  o.nblocks = op->nblocks;
  for (int b = 0; b < op->nblocks; b++) {
    const prost_hip_op_block& B = op->block[b];
    OpBlockDev& D = o.b[b];
    D.kind = B.kind; D.row = B.row; D.col = B.col; D.nrows = B.nrows; D.ncols = B.ncols; D.nx = B.nx; D.ny = B.ny; D.L = B.L;
    D.val = B.val; D.ptr = B.ptr; D.ind = B.ind; D.val_t = B.val_t; D.ptr_t = B.ptr_t; D.ind_t = B.ind_t;
    D.ids = B.ids; D.pptr = B.pptr; D.rel = B.rel; D.pval = B.pval; D.ids_t = B.ids_t; D.pptr_t = B.pptr_t; D.rel_t = B.rel_t; D.pval_t = B.pval_t;
    D.anchor = B.ids ? B.anchor : nullptr; D.anchor_t = B.ids_t ? B.anchor_t : nullptr;
  }
  return o;
}

// The block table in DEVICE memory.  Passed by value it is ~2 KB of kernel arguments; the kernarg segment lives in host-visible memory
// and the table is walked with dependent scalar loads (block -> kind -> pointers -> dominant-pattern entries), so every wavefront paid
// tens of microseconds of round trips to it (measured on the prox kernels with operator sources: 15 us per launch on a 256^2 problem,
// whatever the size).  device_op returns a device copy of the table for `op`, uploaded once and found again by content (a small cache,
// entries live as long as the library); kernels take the pointer and read the table through the scalar cache / L2.
const FusedOpDev* device_op(const prost_hip_fused_op* op);

}  // namespace prost_hip
