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
  // the dominant pattern of K / K^T, copied out of the table (prost_hip_op_block): its operands are requested before the pattern numbers arrive
  int dom_id, dom_n; int32_t dom_rel[PROST_HIP_OP_DOM_MAX]; double dom_val[PROST_HIP_OP_DOM_MAX];
  int dom_id_t, dom_n_t; int32_t dom_rel_t[PROST_HIP_OP_DOM_MAX]; double dom_val_t[PROST_HIP_OP_DOM_MAX];
};
struct FusedOpDev { int nblocks; OpBlockDev b[PROST_HIP_OP_MAX_BLOCKS]; };

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
// of VEC consecutive operands per lane (element-aligned 16-byte accesses, which gfx950 serves); the seams of a stencil walk the table per
// lane or per row.
template <class T, int VEC> struct OpPack { typedef T V __attribute__((ext_vector_type(VEC), aligned(sizeof(T)))); };
// `len`: number of operand elements behind rhs (the speculative loads of the dominant pattern are clamped into [0, len - VEC]: rows at
// the seams of a stencil have other patterns, their speculative values are dropped)
template <class T, int VEC>
__device__ __forceinline__ void pattern_rows(const uint16_t* __restrict__ ids, const int32_t* __restrict__ pptr, const int32_t* __restrict__ rel,
                                             const T* __restrict__ pval, const T* __restrict__ rhs, size_t r0, T (&sum)[VEC],
                                             int dom_id = -1, int dom_n = 0, const int32_t* dom_rel = nullptr, const double* dom_val = nullptr, size_t len = 0) {
  typedef typename OpPack<T, VEC>::V PV;
  unsigned id[VEC];
#pragma unroll
  for (int j = 0; j < VEC; j++) id[j] = ids[r0 + j];
#pragma unroll
  for (int j = 0; j < VEC; j++) sum[j] = 0;
  bool same = true;
#pragma unroll
  for (int j = 1; j < VEC; j++) same = same && id[j] == id[0];
  if (VEC > 1 && dom_n > 0 && len >= (size_t)VEC) {
    // the operands of the DOMINANT pattern are requested at once (table entries from the kernel arguments: no load), in flight together
    // with the pattern numbers above -- one memory round trip instead of four dependent ones (numbers -> table offsets -> entries -> operands)
    constexpr int kB = 6;
    PV x[kB];
    const long hi = (long)len - VEC;
#pragma unroll
    for (int u = 0; u < kB; u++) {
      long a = (long)r0 + (long)dom_rel[u < dom_n ? u : dom_n - 1];
      a = a < 0 ? 0 : (a > hi ? hi : a);
      x[u] = *reinterpret_cast<const PV*>(rhs + a);
    }
    const bool dom = same && id[0] == (unsigned)dom_id;
    if (__builtin_amdgcn_ballot_w64(!dom) == 0) {
#pragma unroll
      for (int u = 0; u < kB; u++) {
        if (u < dom_n) {
          const T v = (T)dom_val[u];
#pragma unroll
          for (int j = 0; j < VEC; j++) sum[j] += v * x[u][j];
        }
      }
      for (int k0 = kB; k0 < dom_n; k0 += kB) {          // (interior rows: every address is in range)
#pragma unroll
        for (int u = 0; u < kB; u++) x[u] = *reinterpret_cast<const PV*>(rhs + (long)r0 + (long)dom_rel[k0 + u < dom_n ? k0 + u : dom_n - 1]);
#pragma unroll
        for (int u = 0; u < kB; u++) {
          if (k0 + u < dom_n) {
            const T v = (T)dom_val[k0 + u];
#pragma unroll
            for (int j = 0; j < VEC; j++) sum[j] += v * x[u][j];
          }
        }
      }
      return;
    }
  }
  if (VEC > 1) {
    const unsigned id0 = (unsigned)__builtin_amdgcn_readfirstlane((int)id[0]);
    const bool uniform = same && id[0] == id0;
    if (__builtin_amdgcn_ballot_w64(!uniform) == 0) {               // every active lane of the wavefront: one pattern
      // entries in batches of eight: the table entries of a batch first (scalar loads), then its eight operand loads in flight together,
      // then the sums in entry order -- one memory round trip per batch instead of one per entry
      const int32_t b = pptr[id0], e = pptr[id0 + 1];
      for (int32_t k0 = b; k0 < e; k0 += 8) {
        long r[8]; T v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int32_t k = k0 + u < e ? k0 + u : e - 1; r[u] = (long)rel[k]; v[u] = pval[k]; }
        typename OpPack<T, VEC>::V x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = *reinterpret_cast<const typename OpPack<T, VEC>::V*>(rhs + (long)r0 + r[u]);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (k0 + u < e) {
#pragma unroll
            for (int j = 0; j < VEC; j++) sum[j] += v[u] * x[u][j];
          }
        }
      }
      return;
    }
    if (same) {                                                     // one pattern per lane
      const int32_t b = pptr[id[0]], e = pptr[id[0] + 1];
      for (int32_t k0 = b; k0 < e; k0 += 4) {
        long r[4]; T v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int32_t k = k0 + u < e ? k0 + u : e - 1; r[u] = (long)rel[k]; v[u] = pval[k]; }
        typename OpPack<T, VEC>::V x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = *reinterpret_cast<const typename OpPack<T, VEC>::V*>(rhs + (long)r0 + r[u]);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (k0 + u < e) {
#pragma unroll
            for (int j = 0; j < VEC; j++) sum[j] += v[u] * x[u][j];
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    const int32_t b = pptr[id[j]], e = pptr[id[j] + 1];
    T s = 0;
    for (int32_t k = b; k < e; k++) s += pval[k] * rhs[(long)(r0 + j) + (long)rel[k]];
    sum[j] = s;
  }
}

// (lanes_in_step = false: the caller's lanes do not take the 64 VEC consecutive elements of a wavefront step together -- the rows of a
// CSR block are then taken lane by lane, never transposed across the wavefront)
// kv[0..VEC) = (K rhs)_(i .. i+VEC).  VEC > 1: i, every block's row / col / nrows and every gradient block's ny and plane
// size are multiples of VEC (host-checked), so the VEC rows lie in the same blocks, the same component plane and image column.
// w0 = the i of lane 0 (wave-uniform); VEC > 1 callers guarantee that all 64 lanes are active.
template <class T, int VEC>
__device__ __forceinline__ void op_fwd_rows(const FusedOpDev& op, size_t i, size_t w0, const T* __restrict__ t, T (&kv)[VEC], bool lanes_in_step = true) {
#pragma unroll
  for (int j = 0; j < VEC; j++) kv[j] = 0;
  for (int b = 0; b < op.nblocks; b++) {
    const OpBlockDev& B = op.b[b];
    if (i < B.row || i >= B.row + B.nrows) continue;
    const size_t r = i - B.row;
    const T* rhs = t + B.col;
    if (B.kind == PROST_OP_CSR) {
      const bool whole = VEC > 1 && lanes_in_step && w0 >= B.row && w0 + (size_t)kWave * VEC <= B.row + B.nrows;
      T sum[VEC];
      if (B.ids) pattern_rows<T, VEC>(B.ids, B.pptr, B.rel, static_cast<const T*>(B.pval), rhs, r, sum, B.dom_id, B.dom_n, B.dom_rel, B.dom_val, (size_t)B.ncols);
      else csr_contrib<T, VEC>(static_cast<const T*>(B.val), B.ptr, B.ind, rhs, r, w0 - B.row, whole, sum);
#pragma unroll
      for (int j = 0; j < VEC; j++) kv[j] = kv[j] + sum[j];
    } else {
      const unsigned nx = (unsigned)B.nx, ny = (unsigned)B.ny, slice = nx * ny, N = slice * (unsigned)B.L;
      const unsigned r32 = (unsigned)r;
      const unsigned c = r32 / N, idx = r32 - c * N;
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
        if (l < (unsigned)B.L - 1) ldv<T, VEC>(rhs + idx + slice, up);
#pragma unroll
        for (int j = 0; j < VEC; j++) g[j] = l < (unsigned)B.L - 1 ? up[j] - cur[j] : -cur[j];      // Dirichlet (block_gradient3d.cu:73-76)
      }
#pragma unroll
      for (int j = 0; j < VEC; j++) kv[j] = kv[j] + g[j];
    }
  }
}
// v[0..VEC) += (K^T rhs)_(j .. j+VEC), blocks in order (EvalAdjointAdd per block)
template <class T, int VEC>
__device__ __forceinline__ void op_adj_cols(const FusedOpDev& op, size_t jg, size_t w0, const T* __restrict__ t, T (&v)[VEC], bool lanes_in_step = true) {
  for (int b = 0; b < op.nblocks; b++) {
    const OpBlockDev& B = op.b[b];
    if (jg < B.col || jg >= B.col + B.ncols) continue;
    const size_t cidx = jg - B.col;
    const T* rhs = t + B.row;
    if (B.kind == PROST_OP_CSR) {
      const bool whole = VEC > 1 && lanes_in_step && w0 >= B.col && w0 + (size_t)kWave * VEC <= B.col + B.ncols;
      T sum[VEC];
      if (B.ids_t) pattern_rows<T, VEC>(B.ids_t, B.pptr_t, B.rel_t, static_cast<const T*>(B.pval_t), rhs, cidx, sum, B.dom_id_t, B.dom_n_t, B.dom_rel_t, B.dom_val_t, (size_t)B.nrows);
      else csr_contrib<T, VEC>(static_cast<const T*>(B.val_t), B.ptr_t, B.ind_t, rhs, cidx, w0 - B.col, whole, sum);
#pragma unroll
      for (int j = 0; j < VEC; j++) v[j] = v[j] + sum[j];
    } else {
      const unsigned nx = (unsigned)B.nx, ny = (unsigned)B.ny, slice = nx * ny, idx = (unsigned)cidx;
      const size_t N = (size_t)slice * B.L;
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
      if (B.kind == PROST_OP_GRAD3D) {
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
        if (B.kind == PROST_OP_GRAD3D) {
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
    D.dom_id = B.dom_id; D.dom_n = B.ids && B.dom_n > 0 && B.dom_n <= PROST_HIP_OP_DOM_MAX ? B.dom_n : 0;
    D.dom_id_t = B.dom_id_t; D.dom_n_t = B.ids_t && B.dom_n_t > 0 && B.dom_n_t <= PROST_HIP_OP_DOM_MAX ? B.dom_n_t : 0;
    for (int k = 0; k < PROST_HIP_OP_DOM_MAX; k++) { D.dom_rel[k] = B.dom_rel[k]; D.dom_val[k] = B.dom_val[k]; D.dom_rel_t[k] = B.dom_rel_t[k]; D.dom_val_t[k] = B.dom_val_t[k]; }
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
