// prost/linop/block.hpp -- plugin base class of linear-operator blocks.
//
// Same contract as the reference's include/prost/linop/block.hpp:37-83: a block sits at
// (row, col) of the big operator, has nrows x ncols entries and ACCUMULATES into the result
// (EvalLocalAdd: res += K_b rhs, EvalAdjointLocalAdd: res += K_b^T rhs).  The only change is the
// iterator type: ranges are raw HBM pointers [begin, end) instead of
// thrust::device_vector<T>::iterator, and kernels are enqueued on prost::CurrentStream().
//
// MI355X additions (optional to override):
//   EvalLocal / EvalAdjointLocal  non-accumulating variants; the default zero-fills and adds.
//       LinearOperator uses them when a block is the only writer of its rows (columns), which
//       removes the thrust::fill pass of linearoperator.cu:140-147.
//   row_sums / col_sums           bulk versions of row_sum / col_sum (the reference makes one
//       virtual call per row and column, problem.cu:262-287).
//   describe                      lets a backend recognise the block and fuse it.
#ifndef PROST_LINOP_BLOCK_HPP_
#define PROST_LINOP_BLOCK_HPP_
#include "prost/common.hpp"

namespace prost {

struct BlockDesc {
  enum Kind { kNone = 0, kGradient2D, kGradient3D, kSparse } kind = kNone;
  size_t nx = 0, ny = 0, L = 0;
  bool label_first = false;
  // kSparse (valid after Initialize()): K and K^T as CSR in device memory, values of type T
  size_t nnz = 0;
  const void* val = nullptr; const int32_t* ptr = nullptr; const int32_t* ind = nullptr;
  const void* val_t = nullptr; const int32_t* ptr_t = nullptr; const int32_t* ind_t = nullptr;
  // kSparse: L > 0 iff the matrix couples the L channels of ONE pixel -- nrows x (L nrows), row i holds exactly L entries, at
  // columns i + c nrows, c = 0 .. L-1 (e.g. [diag(Ix) diag(Iy)]): val is then also the row-major table w[i L + c]
  size_t pointwise_planes = 0;
  // kSparse: a product that runs from row patterns (BlockSparse: one 16-bit pattern number per row + a table) has no CSR arrays on the
  // device -- ids / ids_t non-null then, and val .. ind / val_t .. ind_t null (each direction on its own)
  const uint16_t* ids = nullptr; const int32_t* pptr = nullptr; const int32_t* rel = nullptr; const void* pval = nullptr;
  const uint16_t* ids_t = nullptr; const int32_t* pptr_t = nullptr; const int32_t* rel_t = nullptr; const void* pval_t = nullptr;
  const int32_t* anchor = nullptr; const int32_t* anchor_t = nullptr;     ///< anchored pattern tables (offsets from anchor[row]); null: from the row number
};

template <typename T>
class Block {
 public:
  Block(size_t row, size_t col, size_t nrows, size_t ncols) : row_(row), col_(col), nrows_(nrows), ncols_(ncols) {}
  virtual ~Block() {}

  virtual void Initialize() {}
  virtual void Release() {}

  /// result[row:row+nrows] += K_b rhs[col:col+ncols]      (block.cu:47-57)
  void EvalAdd(T* result, const T* rhs) { EvalLocalAdd(result + row_, result + row_ + nrows_, rhs + col_, rhs + col_ + ncols_); }
  /// result[col:col+ncols] += K_b^T rhs[row:row+nrows]    (block.cu:59-68)
  void EvalAdjointAdd(T* result, const T* rhs) { EvalAdjointLocalAdd(result + col_, result + col_ + ncols_, rhs + row_, rhs + row_ + nrows_); }
  void Eval(T* result, const T* rhs) { EvalLocal(result + row_, result + row_ + nrows_, rhs + col_, rhs + col_ + ncols_); }
  void EvalAdjoint(T* result, const T* rhs) { EvalAdjointLocal(result + col_, result + col_ + ncols_, rhs + row_, rhs + row_ + nrows_); }

  /// sum_j |K_ij|^alpha of LOCAL row / column (host)
  virtual T row_sum(size_t row, T alpha) const = 0;
  virtual T col_sum(size_t col, T alpha) const = 0;
  /// MI355X addition: true iff EVERY row sum is `row` and EVERY column sum is `col` (stencil blocks: the gradient blocks
  /// report 2 and 4 / 6 for every row and column, block_gradient2d.cu:154-163) -- the preconditioners are then two
  /// constants and no per-entry sweep, host vector or upload is needed
  virtual bool uniform_sums(T alpha_row, T alpha_col, T& row, T& col) const { (void)alpha_row; (void)alpha_col; (void)row; (void)col; return false; }
  virtual void row_sums(T* out, T alpha) const { for (size_t r = 0; r < nrows_; r++) out[r] += row_sum(r, alpha); }
  virtual void col_sums(T* out, T alpha) const { for (size_t c = 0; c < ncols_; c++) out[c] += col_sum(c, alpha); }

  size_t row() const { return row_; }
  size_t col() const { return col_; }
  size_t nrows() const { return nrows_; }
  size_t ncols() const { return ncols_; }

  virtual size_t gpu_mem_amount() const = 0;
  virtual bool describe(BlockDesc&) const { return false; }
  /// MI355X addition.  true: this block's OPERATOR is gradient2d(nx, ny, L) (block_gradient2d.cu:26-139: planar output, forward
  /// differences, zero rows at the far borders) although the block is of another kind -- e.g. the sparse matrix
  /// spmat_gradient2d(nx, ny, L) the reference's examples hand over (example_rof_primal.m:10, :28).  d.kind / nx / ny / L are
  /// filled in.  The block's row / column sums stay its own: a backend that runs stencil kernels for it must take the
  /// preconditioners from the problem, not from the stencil block's constants (block_gradient2d.cu:154-163).
  virtual bool stencil_shape(BlockDesc&) const { return false; }

 protected:
  virtual void EvalLocalAdd(T* res_begin, T* res_end, const T* rhs_begin, const T* rhs_end) = 0;
  virtual void EvalAdjointLocalAdd(T* res_begin, T* res_end, const T* rhs_begin, const T* rhs_end) = 0;
  virtual void EvalLocal(T* res_begin, T* res_end, const T* rhs_begin, const T* rhs_end);
  virtual void EvalAdjointLocal(T* res_begin, T* res_end, const T* rhs_begin, const T* rhs_end);

 private:
  size_t row_, col_, nrows_, ncols_;
};

}  // namespace prost
#endif
