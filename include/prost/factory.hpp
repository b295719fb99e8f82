// prost/factory.hpp -- string-keyed registries that turn problem descriptions into objects
// (reference matlab/+prost/private/factory.hpp/.cpp: get_prox_reg / get_block_reg, custom.cpp).
//
// A plugin registers `name -> factory` exactly like the reference's custom.cpp:11-28; the data
// argument is the prost_value tree of include/prost_c.h instead of an mxArray.
#ifndef PROST_FACTORY_HPP_
#define PROST_FACTORY_HPP_
#include <functional>
#include <map>
#include <string>

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

}  // namespace prost
#endif
