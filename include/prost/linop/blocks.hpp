// prost/linop/blocks.hpp -- the in-tree blocks of the hot path
// (reference: include/prost/linop/block_{gradient2d,gradient3d,sparse,diags,zero}.hpp).
#ifndef PROST_LINOP_BLOCKS_HPP_
#define PROST_LINOP_BLOCKS_HPP_
#include "prost/device_vector.hpp"
#include "prost/linop/block.hpp"

namespace prost {

/// forward differences with Neumann boundary, L channels (block_gradient2d.hpp / .cu:141-204)
template <typename T>
class BlockGradient2D : public Block<T> {
 public:
  BlockGradient2D(size_t row, size_t col, size_t nx, size_t ny, size_t L, bool label_first = false)
      : Block<T>(row, col, nx * ny * L * 2, nx * ny * L), nx_(nx), ny_(ny), L_(L), label_first_(label_first) {}
  virtual T row_sum(size_t, T) const { return 2; }     // block_gradient2d.cu:154-157
  virtual T col_sum(size_t, T) const { return 4; }     // :160-163
  virtual bool uniform_sums(T, T, T& row, T& col) const { row = 2; col = 4; return true; }
  virtual void row_sums(T* out, T alpha) const;
  virtual void col_sums(T* out, T alpha) const;
  virtual size_t gpu_mem_amount() const { return 0; }
  virtual bool describe(BlockDesc& d) const { d.kind = BlockDesc::kGradient2D; d.nx = nx_; d.ny = ny_; d.L = L_; d.label_first = label_first_; return true; }

 protected:
  virtual void EvalLocalAdd(T*, T*, const T*, const T*);
  virtual void EvalAdjointLocalAdd(T*, T*, const T*, const T*);
  virtual void EvalLocal(T*, T*, const T*, const T*);
  virtual void EvalAdjointLocal(T*, T*, const T*, const T*);
  size_t nx_, ny_, L_;
  bool label_first_;
};

/// as 2-D plus the label/z difference with Dirichlet boundary at l = L-1 (block_gradient3d.cu:73-76)
template <typename T>
class BlockGradient3D : public Block<T> {
 public:
  BlockGradient3D(size_t row, size_t col, size_t nx, size_t ny, size_t L, bool label_first = false)
      : Block<T>(row, col, nx * ny * L * 3, nx * ny * L), nx_(nx), ny_(ny), L_(L), label_first_(label_first) {}
  virtual T row_sum(size_t, T) const { return 2; }     // block_gradient3d.cu:165-168
  virtual T col_sum(size_t, T) const { return 6; }     // :171-174
  virtual bool uniform_sums(T, T, T& row, T& col) const { row = 2; col = 6; return true; }
  virtual void row_sums(T* out, T alpha) const;
  virtual void col_sums(T* out, T alpha) const;
  virtual size_t gpu_mem_amount() const { return 0; }
  virtual bool describe(BlockDesc& d) const { d.kind = BlockDesc::kGradient3D; d.nx = nx_; d.ny = ny_; d.L = L_; d.label_first = label_first_; return true; }

 protected:
  virtual void EvalLocalAdd(T*, T*, const T*, const T*);
  virtual void EvalAdjointLocalAdd(T*, T*, const T*, const T*);
  virtual void EvalLocal(T*, T*, const T*, const T*);
  virtual void EvalAdjointLocal(T*, T*, const T*, const T*);
  size_t nx_, ny_, L_;
  bool label_first_;
};

/// general sparse block; K as CSR and K^T as CSR (= the CSC arrays handed in) (block_sparse.hpp)
template <typename T>
class BlockSparse : public Block<T> {
 public:
  /// arrays in MATLAB CSC form: ptr = Jc (ncols+1), ind = Ir (nnz) (block_sparse.cu:34-68)
  static BlockSparse<T>* CreateFromCSC(size_t row, size_t col, int m, int n, int nnz, const std::vector<T>& val,
                                       const std::vector<int32_t>& ptr, const std::vector<int32_t>& ind);
  /// the same, taking the arrays over instead of copying them (a 4096^2 gradient handed over as a matrix: 0.8 GB)
  static BlockSparse<T>* CreateFromCSC(size_t row, size_t col, int m, int n, int nnz, std::vector<T>&& val, std::vector<int32_t>&& ptr,
                                       std::vector<int32_t>&& ind);
  virtual void Initialize();
  virtual void Release();
  virtual T row_sum(size_t row, T alpha) const;
  virtual T col_sum(size_t col, T alpha) const;
  virtual void row_sums(T* out, T alpha) const;              ///< all rows at once, on all host cores (same sums as row_sum)
  virtual void col_sums(T* out, T alpha) const;
  virtual size_t gpu_mem_amount() const;
  /// MI355X addition (on by default): a matrix whose rows repeat a few (column - row, value) sequences -- a stencil written out as a
  /// sparse matrix -- is applied from one 16-bit pattern number per row + a small table instead of its CSR arrays; same sums, same order
  static void SetPatternCompression(bool on);
  static bool pattern_compression();
  /// which of K / K^T are applied from row patterns (after Initialize()), and how many patterns each has
  bool patterns_forward() const { return pat_.on; }
  bool patterns_adjoint() const { return pat_t_.on; }
  size_t pattern_count(bool adjoint) const { return (adjoint ? pat_t_ : pat_).count; }
  /// the matrix IS spmat_gradient2d(nx, ny, L) (matlab/+prost/+test/private/spmat_gradient2d.m:7-14), entry for entry (Initialize())
  virtual bool stencil_shape(BlockDesc& d) const {
    if (!grad_nx_) return false;
    d.kind = BlockDesc::kGradient2D; d.nx = grad_nx_; d.ny = grad_ny_; d.L = grad_L_; d.label_first = false;
    return true;
  }
  static void SetStencilRecognition(bool on);
  virtual bool describe(BlockDesc& d) const {
    if (nnz_ == 0) return false;
    if (!pat_.on && val_.size() != nnz_) return false;          // before Initialize()
    if (!pat_t_.on && val_t_.size() != nnz_) return false;
    d.kind = BlockDesc::kSparse; d.nnz = nnz_;
    if (pat_.on) { d.ids = pat_.ids.data(); d.pptr = pat_.pptr.data(); d.rel = pat_.rel.data(); d.pval = pat_.val.data(); d.anchor = pat_.anchor.size() ? pat_.anchor.data() : nullptr; }
    else { d.val = val_.data(); d.ptr = ptr_.data(); d.ind = ind_.data(); }
    if (pat_t_.on) { d.ids_t = pat_t_.ids.data(); d.pptr_t = pat_t_.pptr.data(); d.rel_t = pat_t_.rel.data(); d.pval_t = pat_t_.val.data(); d.anchor_t = pat_t_.anchor.size() ? pat_t_.anchor.data() : nullptr; }
    else { d.val_t = val_t_.data(); d.ptr_t = ptr_t_.data(); d.ind_t = ind_t_.data(); }
    d.pointwise_planes = pointwise_planes_;
    return true;
  }

 protected:
  BlockSparse(size_t row, size_t col, size_t nrows, size_t ncols) : Block<T>(row, col, nrows, ncols), nnz_(0) {}
  virtual void EvalLocalAdd(T*, T*, const T*, const T*);
  virtual void EvalAdjointLocalAdd(T*, T*, const T*, const T*);
  virtual void EvalLocal(T*, T*, const T*, const T*);            ///< non-accumulating product in one pass (no separate zero fill)
  virtual void EvalAdjointLocal(T*, T*, const T*, const T*);
  size_t nnz_;
  size_t grad_nx_ = 0, grad_ny_ = 0, grad_L_ = 0;      ///< non-zero: K == spmat_gradient2d(grad_nx_, grad_ny_, grad_L_)
  void DetectGradient2D();
  size_t pointwise_planes_ = 0;                        ///< non-zero: row i has its entries at columns i + c nrows, c < pointwise_planes_ (Initialize())
  void DetectPointwise();
  std::vector<int32_t> host_ind_, host_ind_t_, host_ptr_, host_ptr_t_;
  std::vector<T> host_val_, host_val_t_;
  device_vector<int32_t> ind_, ind_t_, ptr_, ptr_t_;
  device_vector<T> val_, val_t_;
  struct RowPatterns {
    bool on = false;
    size_t count = 0;
    device_vector<uint16_t> ids;              ///< pattern number of every row
    device_vector<int32_t> pptr, rel;         ///< entries pptr[id] .. pptr[id + 1] - 1 of the table: column - row ...
    device_vector<T> val;                     ///< ... and value
    device_vector<int32_t> anchor;            ///< anchored table (matrices between different geometries): offsets count from anchor[row]; empty: from the row number
  };
  RowPatterns pat_, pat_t_;
  void PatternProduct(const RowPatterns& p, T* r, const T* x, size_t rows, int acc);
};

/// kron(K, I_d) (id_first == false, block_sparse_kron_id.cu) or kron(I_d, K) (id_first == true,
/// block_id_kron_sparse.cu) for a small sparse K, without forming the product; values kept as float
template <typename T>
class BlockKronSparse : public Block<T> {
 public:
  static BlockKronSparse<T>* CreateFromCSC(bool id_first, size_t row, size_t col, size_t diaglength, int m, int n, int nnz,
                                           const std::vector<T>& val, const std::vector<int32_t>& ptr, const std::vector<int32_t>& ind);
  virtual void Initialize();
  virtual void Release();
  virtual T row_sum(size_t row, T alpha) const;
  virtual T col_sum(size_t col, T alpha) const;
  virtual size_t gpu_mem_amount() const;

 protected:
  BlockKronSparse(size_t row, size_t col, size_t nrows, size_t ncols) : Block<T>(row, col, nrows, ncols) {}
  virtual void EvalLocalAdd(T*, T*, const T*, const T*);
  virtual void EvalAdjointLocalAdd(T*, T*, const T*, const T*);
  virtual void EvalLocal(T*, T*, const T*, const T*);              ///< non-accumulating forms: no fill pass in front of the product
  virtual void EvalAdjointLocal(T*, T*, const T*, const T*);
  bool id_first_ = false;
  size_t diaglength_ = 0, mat_nnz_ = 0, mat_nrows_ = 0, mat_ncols_ = 0;
  std::vector<int32_t> host_ind_, host_ind_t_, host_ptr_, host_ptr_t_;
  std::vector<float> host_val_, host_val_t_;
  device_vector<int32_t> ind_, ind_t_, ptr_, ptr_t_;
  device_vector<float> val_, val_t_;
};

/// constant-coefficient multi-diagonal block; `identity` maps here (block_diags.hpp, identity.m:11-12)
template <typename T>
class BlockDiags : public Block<T> {
 public:
  BlockDiags(size_t row, size_t col, size_t nrows, size_t ncols, size_t ndiags, const std::vector<int64_t>& offsets,
             const std::vector<T>& factors);
  virtual void Initialize();
  virtual void Release();
  virtual T row_sum(size_t row, T alpha) const;
  virtual T col_sum(size_t col, T alpha) const;
  virtual size_t gpu_mem_amount() const { return ndiags_ * (sizeof(float) + sizeof(int64_t)); }
  /// reproduce the reference's adjoint launch grid sized from nrows (block_diags.cu:211)?
  /// default false: every column is written.
  static void SetReferenceGridQuirk(bool on);
  static bool ReferenceGridQuirk();

 protected:
  virtual void EvalLocalAdd(T*, T*, const T*, const T*);
  virtual void EvalAdjointLocalAdd(T*, T*, const T*, const T*);
  size_t ndiags_;
  std::vector<int64_t> offsets_;
  std::vector<float> factors_;      // float regardless of T (block_diags.cu:30,:108)
  device_vector<int64_t> d_offsets_;
  device_vector<float> d_factors_;
};

template <typename T>
class BlockZero : public Block<T> {
 public:
  BlockZero(size_t row, size_t col, size_t nrows, size_t ncols) : Block<T>(row, col, nrows, ncols) {}
  virtual T row_sum(size_t, T) const { return 0; }
  virtual T col_sum(size_t, T) const { return 0; }
  virtual size_t gpu_mem_amount() const { return 0; }

 protected:
  virtual void EvalLocalAdd(T*, T*, const T*, const T*) {}
  virtual void EvalAdjointLocalAdd(T*, T*, const T*, const T*) {}
};

}  // namespace prost
#endif
