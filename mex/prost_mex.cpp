// prost_mex.cpp -- the MATLAB MEX gateway over libprost.so.
//
// Replaces matlab/+prost/private/prost.cpp of the reference: the MATLAB package (matlab/+prost/*.m) keeps calling
//     prost_('solve_problem', prob.data, nrows, ncols, backend, opts)   etc.
// and this mexFunction forwards command and arguments to prost_command() (include/prost_c.h), whose command table is the
// reference's (prost.cpp:305-313).  The work the reference's gateway did itself -- factories, Solver, device reset -- lives
// behind that call; what is left here is the mxArray <-> prost_value marshalling:
//     convert()   mxArray -> prost_value  : numeric / logical matrices, char arrays, cells, structs, sparse matrices
//                                           (mxGetIr / mxGetJc as factory.cpp:633-645 reads them), function handles
//                                           (the intermediate-solution callback, factory.cpp:136-158)
//     back()      prost_value -> mxArray  : matrices, strings, cells and the {x, y, z, w, result} struct of prost.cpp:138-152
//     stop_cb()   Ctrl-C polling          : prost.cpp:58-66
//     print_cb()  library output          : what libprost.so prints (verbose header, "It k: Feas_p=..." lines, list_gpus) arrives
//                                           through prost_set_output_callback and goes to mexPrintf, with the pause(.001) after
//                                           a newline that lets MATLAB flush its command window (prost.cpp:15-44 mexstream /
//                                           scoped_redirect_cout); installed for the duration of one mexFunction call
// Errors: prost_command returns non-zero, the message goes to mexErrMsgTxt (prost.cpp:342-346).
//
// Build (replaces matlab/CMakeLists.txt:46-57):  mex mex/prost_mex.cpp -Iinclude -Lprost_amd/lib -lprost -output prost_
// MATLAB and mex.h do not exist in the build image, so this file cannot be linked here; tests/test_frontend.py keeps it
// compiling (g++ -fsyntax-only against tests/mex_decl.h, declarations of the MEX API it uses -- test infrastructure).
#ifndef PROST_MEX_DECLARATIONS_PROVIDED
#include <mex.h>
#endif

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "prost_c.h"

// undocumented but stable libut entry points the reference polls for Ctrl-C (prost.cpp:46-52)
extern "C" bool utIsInterruptPending();
extern "C" void utSetInterruptPending(bool);

namespace {

int stop_cb(void*) {                                         // prost.cpp:58-66 MexStoppingCallback
  if (utIsInterruptPending()) {
    utSetInterruptPending(false);
    return 1;
  }
  return 0;
}

// prost.cpp:15-33 mexstream: text to the MATLAB command window; after a newline MATLAB gets the chance to flush it
void print_cb(void*, const char* text, size_t n) {
  if (n == 0) return;
  mexPrintf("%.*s", static_cast<int>(n), text);
  if (text[n - 1] == '\n') mexEvalString("pause(.001);");
}

struct ScopedOutputRedirect {            // prost.cpp:35-44 scoped_redirect_cout
  ScopedOutputRedirect() { prost_set_output_callback(print_cb, nullptr); }
  ~ScopedOutputRedirect() { prost_set_output_callback(nullptr, nullptr); }
};

// factory.cpp:136-158 SolverIntermCallback: feval(handle, iter, primal, dual) -> is_converged
int interm_cb(void* user, int iteration, const double* x, size_t nx, const double* y, size_t ny) {
  mxArray* rhs[4];
  rhs[0] = static_cast<mxArray*>(user);
  rhs[1] = mxCreateDoubleScalar(iteration);
  rhs[2] = mxCreateDoubleMatrix(nx, 1, mxREAL);
  rhs[3] = mxCreateDoubleMatrix(ny, 1, mxREAL);
  std::copy(x, x + nx, mxGetPr(rhs[2]));
  std::copy(y, y + ny, mxGetPr(rhs[3]));
  mxArray* lhs[1] = {nullptr};
  mexCallMATLAB(1, lhs, 4, rhs, "feval");
  mxDestroyArray(rhs[1]);
  mxDestroyArray(rhs[2]);
  mxDestroyArray(rhs[3]);
  const bool converged = lhs[0] && mxGetScalar(lhs[0]) != 0;
  if (lhs[0]) mxDestroyArray(lhs[0]);
  return converged ? 1 : 0;
}

// numeric classes other than double are widened; the library narrows to its `real` itself (factory.cpp:161-283)
template <typename S>
prost_value* widen(const mxArray* a) {
  const size_t n = mxGetNumberOfElements(a);
  const S* p = static_cast<const S*>(mxGetData(a));
  std::vector<double> d(p, p + n);
  return prost_value_matrix(d.data(), mxGetM(a), mxGetN(a));
}

prost_value* convert(const mxArray* a) {
  if (!a) return prost_value_matrix(nullptr, 0, 0);
  if (mxIsCell(a)) {
    const size_t n = mxGetNumberOfElements(a);
    prost_value* c = prost_value_cell(n);
    for (size_t i = 0; i < n; i++) prost_value_cell_set(c, i, convert(mxGetCell(a, i)));
    return c;
  }
  if (mxIsStruct(a)) {                                       // the problem / options structs: 1 x 1 (factory.cpp:950-1012)
    prost_value* s = prost_value_struct();
    const int nf = mxGetNumberOfFields(a);
    if (mxGetNumberOfElements(a) > 0)
      for (int f = 0; f < nf; f++) prost_value_struct_set(s, mxGetFieldNameByNumber(a, f), convert(mxGetFieldByNumber(a, 0, f)));
    return s;
  }
  if (mxIsClass(a, "function_handle")) return prost_value_callback(interm_cb, const_cast<mxArray*>(a));
  if (mxIsChar(a)) {
    char* str = mxArrayToString(a);
    prost_value* v = prost_value_string(str ? str : "");
    if (str) mxFree(str);
    return v;
  }
  if (mxIsSparse(a)) {                                       // MATLAB stores CSC: values, row indices, column starts
    const size_t ncols = mxGetN(a);
    const mwIndex* ir = mxGetIr(a);
    const mwIndex* jc = mxGetJc(a);
    const size_t nnz = static_cast<size_t>(jc[ncols]);
    std::vector<int64_t> ir64(ir, ir + nnz), jc64(jc, jc + ncols + 1);
    return prost_value_sparse(mxGetM(a), ncols, nnz, mxGetPr(a), ir64.data(), jc64.data());
  }
  if (mxIsEmpty(a)) return prost_value_matrix(nullptr, mxGetM(a), mxGetN(a));
  if (mxIsLogical(a)) return widen<mxLogical>(a);
  if (mxIsDouble(a)) return prost_value_matrix(mxGetPr(a), mxGetM(a), mxGetN(a));
  if (mxIsSingle(a)) return widen<float>(a);
  if (mxIsInt32(a)) return widen<int32_t>(a);
  if (mxIsUint32(a)) return widen<uint32_t>(a);
  if (mxIsInt64(a)) return widen<int64_t>(a);
  if (mxIsUint64(a)) return widen<uint64_t>(a);
  if (mxIsInt8(a)) return widen<int8_t>(a);
  if (mxIsUint8(a)) return widen<uint8_t>(a);
  if (mxIsInt16(a)) return widen<int16_t>(a);
  if (mxIsUint16(a)) return widen<uint16_t>(a);
  mexErrMsgTxt("prost_: unsupported argument class.");
  return nullptr;
}

// Results.  solve_problem returns the struct of prost.cpp:138-152 ({x, y, z, w, result}; this library adds iters, path,
// pair_launches); every struct a command returns is copied field by field, whatever the command put into it
// (prost_value_field_count / prost_value_field_name).
mxArray* back(const prost_value* v) {
  switch (prost_value_kind(v)) {
    case PROST_VALUE_MATRIX: {
      const size_t r = prost_value_rows(v), c = prost_value_cols(v);
      mxArray* m = mxCreateDoubleMatrix(r, c, mxREAL);
      const double* d = prost_value_data(v);
      if (r * c > 0) std::copy(d, d + r * c, mxGetPr(m));
      return m;
    }
    case PROST_VALUE_STRING:
      return mxCreateString(prost_value_str(v));
    case PROST_VALUE_CELL: {
      const size_t n = prost_value_count(v);
      mxArray* c = mxCreateCellMatrix(n, n ? 1 : 0);
      for (size_t i = 0; i < n; i++) mxSetCell(c, i, back(prost_value_cell_get(v, i)));
      return c;
    }
    case PROST_VALUE_STRUCT: {
      std::vector<const char*> names;
      for (size_t i = 0; i < prost_value_field_count(v); i++) names.push_back(prost_value_field_name(v, i));
      mxArray* s = mxCreateStructMatrix(1, 1, static_cast<int>(names.size()), names.data());
      for (size_t i = 0; i < names.size(); i++) mxSetFieldByNumber(s, 0, static_cast<int>(i), back(prost_value_field(v, names[i])));
      return s;
    }
    default:
      return mxCreateDoubleMatrix(0, 0, mxREAL);
  }
}

struct ValueList {                       // frees the converted arguments on every exit path short of mexErrMsgTxt's longjmp
  std::vector<prost_value*> v;
  ~ValueList() { for (prost_value* p : v) prost_value_free(p); }
};

}  // namespace

void mexFunction(int nlhs, mxArray** plhs, int nrhs, const mxArray** prhs) {
  if (nrhs == 0) mexErrMsgTxt("Usage: prost_(command, arg1, arg2, ...);");                    // prost.cpp:317-318
  char* cmd_chars = mxArrayToString(prhs[0]);
  const std::string cmd = cmd_chars ? cmd_chars : "";
  if (cmd_chars) mxFree(cmd_chars);

  std::string error;
  const int nout = std::max(nlhs, 1);
  {
    ValueList in, out;
    ScopedOutputRedirect redirect;                            // the library's std::cout -> mexPrintf while the command runs
    for (int i = 1; i < nrhs; i++) in.v.push_back(convert(prhs[i]));
    out.v.assign(static_cast<size_t>(nout), nullptr);
    prost_set_stop_callback(stop_cb, nullptr);
    const int rc = prost_command(cmd.c_str(), nout, out.v.data(), static_cast<int>(in.v.size()), in.v.data());
    if (rc != 0) error = prost_last_error();
    else for (int i = 0; i < nlhs; i++) plhs[i] = back(out.v[static_cast<size_t>(i)]);
  }                                                           // prost_values released here, before any longjmp
  if (!error.empty()) {
    prost_value* none = nullptr;
    prost_command("release", 0, &none, 0, nullptr);           // prost.cpp:344-345: release after an error
    mexErrMsgTxt(error.c_str());
  }
}
