// prost/factory.hpp -- string-keyed registries that turn problem descriptions into objects
// (reference matlab/+prost/private/factory.hpp/.cpp: get_prox_reg / get_block_reg, custom.cpp).
//
// A plugin registers `name -> factory` exactly like the reference's custom.cpp:11-28; the data
// argument is the prost_value tree of include/prost_c.h instead of an mxArray.
#ifndef PROST_FACTORY_HPP_
#define PROST_FACTORY_HPP_
#include <array>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "prost/prost.hpp"
#include "prost_c.h"

namespace prost {

template <typename T>
struct Factory {
  typedef std::function<Prox<T>*(size_t idx, size_t size, bool diagsteps, const prost_value* data)> ProxFactory;
  typedef std::function<Block<T>*(size_t row, size_t col, const prost_value* data)> BlockFactory;
  typedef std::function<Backend<T>*(const prost_value* opts)> BackendFactory;

  static std::map<std::string, ProxFactory>& prox_reg();
  static std::map<std::string, BlockFactory>& block_reg();
  static std::map<std::string, BackendFactory>& backend_reg();

  /// prox cell {name, idx, size, diagsteps, data}           (factory.cpp:820-867)
  static shared_ptr<Prox<T>> CreateProx(const prost_value* cell);
  /// block cell {name, row, col, data}                      (factory.cpp:869-912)
  static shared_ptr<Block<T>> CreateBlock(const prost_value* cell);
  /// backend cell {name, optsStruct}, name lower-cased      (factory.cpp:914-948)
  static shared_ptr<Backend<T>> CreateBackend(const prost_value* cell);
  /// problem struct                                         (factory.cpp:950-990)
  static shared_ptr<Problem<T>> CreateProblem(const prost_value* s, size_t nrows, size_t ncols);
  /// opts struct; interm_cb returned separately             (factory.cpp:992-1012)
  static typename Solver<T>::Options CreateSolverOptions(const prost_value* s);
};

// helpers for plugin authors (factory.cpp:161-283)
std::vector<double> GetVector(const prost_value* v);
double GetScalarFromCell(const prost_value* cell, size_t index);
double GetScalarFromField(const prost_value* s, const std::string& name);
std::string GetString(const prost_value* v);
/// element `index` of a cell array (mxGetCell with the bounds check of factory.cpp:238-246)
inline const prost_value* GetCell(const prost_value* cell, size_t index) {
  if (!cell || prost_value_kind(cell) != PROST_VALUE_CELL || index >= prost_value_count(cell) || !prost_value_cell_get(cell, index))
    throw Exception("Out-of-bounds access into cell-array.");
  return prost_value_cell_get(cell, index);
}
/// a cell array of COEFFS_COUNT vectors -> std::array of std::vector<T>, every entry 1 or `count` values (factory.cpp:196-216):
/// the coefficient argument of ProxElemOperation<T, ELEM_OPERATION> (prox_elem_operation.hpp)
template <typename T, size_t COEFFS_COUNT>
void GetCoefficients(std::array<std::vector<T>, COEFFS_COUNT>& coeffs, const prost_value* cell_arr, size_t count) {
  if (!cell_arr || prost_value_kind(cell_arr) != PROST_VALUE_CELL || prost_value_count(cell_arr) < COEFFS_COUNT)
    throw Exception("Cell array of coefficients is too small.");
  for (size_t i = 0; i < COEFFS_COUNT; i++) {
    const std::vector<double> v = GetVector(prost_value_cell_get(cell_arr, i));
    coeffs[i].assign(v.begin(), v.end());
    if (coeffs[i].size() != 1 && coeffs[i].size() != count) throw Exception("Size of coefficients should be either 1 or count.");
  }
}

}  // namespace prost
#endif
