// prost/linop/linearoperator.hpp -- block container (reference linearoperator.hpp:36-90,
// dual_linearoperator.hpp).
#ifndef PROST_LINOP_LINEAROPERATOR_HPP_
#define PROST_LINOP_LINEAROPERATOR_HPP_
#include "prost/device_vector.hpp"
#include "prost/linop/block.hpp"

namespace prost {

template <typename T>
class LinearOperator {
 public:
  LinearOperator() : nrows_(0), ncols_(0), rows_exclusive_(false), cols_exclusive_(false) {}
  virtual ~LinearOperator() {}

  void AddBlock(shared_ptr<Block<T>> block) { blocks_.push_back(block); }
  /// overlap check + sizes (host only; linearoperator.cu:84-120)
  virtual void InitializeHost();
  /// InitializeHost + Block::Initialize uploads (linearoperator.cu:122-125)
  virtual void Initialize();
  virtual void Release();

  /// result = beta * result + K rhs   (linearoperator.cu:135-151)
  virtual void Eval(device_vector<T>& result, const device_vector<T>& rhs, T beta = 0);
  /// result = beta * result + K^T rhs (linearoperator.cu:153-170)
  virtual void EvalAdjoint(device_vector<T>& result, const device_vector<T>& rhs, T beta = 0);
  /// host-vector versions; return the mean milliseconds of 5 repeats (linearoperator.cu:173-220)
  double Eval(std::vector<T>& result, const std::vector<T>& rhs);
  double EvalAdjoint(std::vector<T>& result, const std::vector<T>& rhs);

  virtual T row_sum(size_t row, T alpha) const;
  virtual T col_sum(size_t col, T alpha) const;
  /// bulk: out[r] = sum over blocks; out must hold nrows() / ncols() entries
  virtual void row_sums(std::vector<T>& out, T alpha) const;
  virtual void col_sums(std::vector<T>& out, T alpha) const;
  /// true iff the operator is ONE block that spans it and reports constant row and column sums (Block::uniform_sums)
  virtual bool uniform_sums(T alpha_row, T alpha_col, T& row, T& col) const {
    if (blocks_.size() != 1) return false;
    const auto& b = blocks_[0];
    return b->row() == 0 && b->col() == 0 && b->nrows() == nrows_ && b->ncols() == ncols_ && b->uniform_sums(alpha_row, alpha_col, row, col);
  }

  virtual size_t nrows() const { return nrows_; }
  virtual size_t ncols() const { return ncols_; }
  virtual size_t gpu_mem_amount() const;
  const std::vector<shared_ptr<Block<T>>>& blocks() const { return blocks_; }

 protected:
  void ApplyBeta(device_vector<T>& result, T beta, bool negate_beta);
  std::vector<shared_ptr<Block<T>>> blocks_;
  size_t nrows_, ncols_;
  bool rows_exclusive_, cols_exclusive_;   // every row (column) written by exactly one block
  /// MI355X addition: per block 0 = the first writer of its whole row (column) range, 1 = every row (column) of its range has been written by
  /// earlier blocks; empty = no such plan (a block's range is partly new, or some row has no writer): fill + accumulate as the reference does
  std::vector<char> row_plan_, col_plan_;

  template <typename U> friend class DualLinearOperator;
};

/// -K^T view used by Problem::Dualize (dual_linearoperator.cu:39-80)
template <typename T>
class DualLinearOperator : public LinearOperator<T> {
 public:
  explicit DualLinearOperator(shared_ptr<LinearOperator<T>> child) : child_(child) {}
  virtual void InitializeHost() {}
  virtual void Initialize() {}
  virtual void Release() {}
  virtual void Eval(device_vector<T>& result, const device_vector<T>& rhs, T beta = 0);
  virtual void EvalAdjoint(device_vector<T>& result, const device_vector<T>& rhs, T beta = 0);
  virtual T row_sum(size_t row, T alpha) const { return child_->col_sum(row, alpha); }
  virtual T col_sum(size_t col, T alpha) const { return child_->row_sum(col, alpha); }
  virtual void row_sums(std::vector<T>& out, T alpha) const { child_->col_sums(out, alpha); }
  virtual void col_sums(std::vector<T>& out, T alpha) const { child_->row_sums(out, alpha); }
  virtual bool uniform_sums(T alpha_row, T alpha_col, T& row, T& col) const { return child_->uniform_sums(alpha_col, alpha_row, col, row); }
  virtual size_t nrows() const { return child_->ncols(); }
  virtual size_t ncols() const { return child_->nrows(); }
  virtual size_t gpu_mem_amount() const { return 0; }
  /// thrust::negate<float> round trip for T = double (dual_linearoperator.cu:56-57); default off
  static void SetReferenceNegateQuirk(bool on);

 protected:
  shared_ptr<LinearOperator<T>> child_;
};

}  // namespace prost
#endif
