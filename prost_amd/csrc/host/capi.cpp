// capi.cpp -- prost_value tree, factories/registries and the command table of include/prost_c.h
// (the MEX gateway of the reference, matlab/+prost/private/{prost,factory}.cpp, without mex.h).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <iostream>
#include <map>
#include <sstream>

#include "hipapi.hpp"
#include "prost/factory.hpp"

using namespace prost;

// ------------------------------------------------------------------------------------------
// prost_value
// ------------------------------------------------------------------------------------------
// value storage without the zero fill of std::vector::resize: the 10^7..10^8-element result vectors are written once, by
// several host threads (vec_value_t)
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U> struct rebind { typedef default_init_allocator<U> other; };
  using std::allocator<T>::allocator;
  template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
typedef std::vector<double, default_init_allocator<double>> value_vector;

// std::cout of the library (the verbose header and "It k: Feas_p=..." lines of Solver::Solve, list_gpus, the |K| rescale
// note) handed to a front end's print function: the mexstream of prost.cpp:15-44 behind the C ABI
namespace {
prost_output_cb g_out_fn = nullptr;
void* g_out_user = nullptr;
std::streambuf* g_cout_saved = nullptr;
class CallbackStreambuf : public std::streambuf {
 protected:
  std::streamsize xsputn(const char* s, std::streamsize n) override { if (g_out_fn && n > 0) g_out_fn(g_out_user, s, (size_t)n); return n; }
  int overflow(int c = EOF) override {
    if (c != EOF) { const char ch = (char)c; if (g_out_fn) g_out_fn(g_out_user, &ch, 1); }
    return 1;
  }
};
CallbackStreambuf g_out_buf;
}  // namespace

struct prost_value {
  int kind = PROST_VALUE_EMPTY;
  size_t rows = 0, cols = 0;
  value_vector data;                    // matrix values / sparse values
  std::vector<int64_t> ir, jc;          // sparse
  std::string str;
  std::vector<prost_value*> cells;
  std::vector<std::pair<std::string, prost_value*>> fields;
  prost_interm_cb cb = nullptr;
  void* cb_user = nullptr;
  ~prost_value() {
    for (auto* c : cells) delete c;
    for (auto& f : fields) delete f.second;
  }
};

static thread_local std::string g_error;
static prost_stop_cb g_stop_cb = nullptr;
static void* g_stop_user = nullptr;

extern "C" {

prost_value* prost_value_scalar(double v) { return prost_value_matrix(&v, 1, 1); }
prost_value* prost_value_matrix(const double* data, size_t rows, size_t cols) {
  prost_value* v = new prost_value; v->kind = PROST_VALUE_MATRIX; v->rows = rows; v->cols = cols;
  const size_t n = rows * cols;
  if (data && n >= ((size_t)1 << 20)) {       // large vectors (per-pixel coefficients): storage not zero-filled, copied on several host threads
    v->data.resize(n);
    double* dst = v->data.data();
    prost::ParallelFor(n, [&](size_t b, size_t e) { std::memcpy(dst + b, data + b, (e - b) * sizeof(double)); });
  } else if (data && n > 0) v->data.assign(data, data + n);
  else v->data.assign(n, 0.0);
  return v;
}
prost_value* prost_value_string(const char* s) { prost_value* v = new prost_value; v->kind = PROST_VALUE_STRING; v->str = s ? s : ""; v->rows = 1; v->cols = v->str.size(); return v; }
prost_value* prost_value_cell(size_t n) { prost_value* v = new prost_value; v->kind = PROST_VALUE_CELL; v->cells.assign(n, nullptr); v->rows = n; v->cols = n ? 1 : 0; return v; }
int prost_value_cell_set(prost_value* cell, size_t i, prost_value* v) {
  if (!cell || cell->kind != PROST_VALUE_CELL || i >= cell->cells.size()) return 1;
  delete cell->cells[i]; cell->cells[i] = v; return 0;
}
prost_value* prost_value_struct(void) { prost_value* v = new prost_value; v->kind = PROST_VALUE_STRUCT; v->rows = v->cols = 1; return v; }
int prost_value_struct_set(prost_value* s, const char* name, prost_value* v) {
  if (!s || s->kind != PROST_VALUE_STRUCT) return 1;
  for (auto& f : s->fields) if (f.first == name) { delete f.second; f.second = v; return 0; }
  s->fields.emplace_back(name, v); return 0;
}
prost_value* prost_value_sparse(size_t rows, size_t cols, size_t nnz, const double* val, const int64_t* ir, const int64_t* jc) {
  prost_value* v = new prost_value; v->kind = PROST_VALUE_SPARSE; v->rows = rows; v->cols = cols;
  v->data.assign(val, val + nnz); v->ir.assign(ir, ir + nnz); v->jc.assign(jc, jc + cols + 1);
  return v;
}
prost_value* prost_value_callback(prost_interm_cb fn, void* user) { prost_value* v = new prost_value; v->kind = PROST_VALUE_CALLBACK; v->cb = fn; v->cb_user = user; return v; }
void prost_value_free(prost_value* v) { delete v; }

int prost_value_kind(const prost_value* v) { return v ? v->kind : PROST_VALUE_EMPTY; }
size_t prost_value_rows(const prost_value* v) { return v ? v->rows : 0; }
size_t prost_value_cols(const prost_value* v) { return v ? v->cols : 0; }
const double* prost_value_data(const prost_value* v) { return v ? v->data.data() : nullptr; }
const char* prost_value_str(const prost_value* v) { return v ? v->str.c_str() : ""; }
size_t prost_value_count(const prost_value* v) { return v ? v->cells.size() : 0; }
const prost_value* prost_value_cell_get(const prost_value* v, size_t i) { return (v && i < v->cells.size()) ? v->cells[i] : nullptr; }
const prost_value* prost_value_field(const prost_value* v, const char* name) {
  if (!v || v->kind != PROST_VALUE_STRUCT) return nullptr;
  for (auto& f : v->fields) if (f.first == name) return f.second;
  return nullptr;
}
size_t prost_value_field_count(const prost_value* v) { return (v && v->kind == PROST_VALUE_STRUCT) ? v->fields.size() : 0; }
const char* prost_value_field_name(const prost_value* v, size_t i) {
  return (v && v->kind == PROST_VALUE_STRUCT && i < v->fields.size()) ? v->fields[i].first.c_str() : nullptr;
}
const char* prost_last_error(void) { return g_error.c_str(); }
void prost_set_output_callback(prost_output_cb fn, void* user) {
  g_out_fn = fn; g_out_user = user;
  if (fn) { if (!g_cout_saved) g_cout_saved = std::cout.rdbuf(&g_out_buf); }
  else if (g_cout_saved) { std::cout.rdbuf(g_cout_saved); g_cout_saved = nullptr; }
}
void prost_set_stop_callback(prost_stop_cb fn, void* user) { g_stop_cb = fn; g_stop_user = user; }

}  // extern "C"

// ------------------------------------------------------------------------------------------
// description helpers (factory.cpp:161-283)
// ------------------------------------------------------------------------------------------
namespace prost {

std::vector<double> GetVector(const prost_value* v) {
  if (!v || v->kind != PROST_VALUE_MATRIX) throw Exception("Argument has to be passed as a vector of type single or double.");
  if (v->cols != 1 && v->rows != 1) throw Exception("Vector has to be Nx1 or 1xN.");
  if (v->rows == 0 || v->cols == 0) throw Exception("Empty vector passed.");
  return std::vector<double>(v->data.begin(), v->data.end());
}
static double scalar_of(const prost_value* v) {
  if (!v || v->kind != PROST_VALUE_MATRIX || v->data.empty()) throw Exception("Scalar expected.");
  return v->data[0];
}
double GetScalarFromCell(const prost_value* cell, size_t index) {
  if (!cell || cell->kind != PROST_VALUE_CELL || index >= cell->cells.size()) throw Exception("Out-of-bounds access into cell-array.");
  return scalar_of(cell->cells[index]);
}
double GetScalarFromField(const prost_value* s, const std::string& name) {
  const prost_value* f = prost_value_field(s, name.c_str());
  if (!f) { std::stringstream ss; ss << "Field with name '" << name << "' not found."; throw Exception(ss.str()); }
  return scalar_of(f);
}
std::string GetString(const prost_value* v) {
  if (!v || v->kind != PROST_VALUE_STRING) throw Exception("String expected.");
  return v->str;
}
static const prost_value* cell_at(const prost_value* c, size_t i) {
  if (!c || c->kind != PROST_VALUE_CELL || i >= c->cells.size() || !c->cells[i]) throw Exception("Out-of-bounds access into cell-array.");
  return c->cells[i];
}
static std::vector<const prost_value*> cell_list(const prost_value* c) {
  std::vector<const prost_value*> out;
  if (!c) throw Exception("Tried to run GetCellArray on non-existing array.");
  if (c->kind == PROST_VALUE_EMPTY || (c->kind == PROST_VALUE_MATRIX && c->data.empty())) return out;
  if (c->kind != PROST_VALUE_CELL) throw Exception("Cell array expected.");
  for (auto* e : c->cells) out.push_back(e);
  return out;
}

static const char* kFunctionNames[PROST_FN_COUNT] = {"zero", "abs", "square", "ind_leq0", "ind_geq0", "ind_eq0", "ind_box01",
                                                     "max_pos0", "l0", "huber", "lq", "lq_plus_eps", "trunclin", "truncquad"};

template <typename T>
static void get_coefficients(std::array<std::vector<T>, 7>& coeffs, const prost_value* cell_arr, size_t count) {
  if (!cell_arr || cell_arr->kind != PROST_VALUE_CELL || cell_arr->cells.size() < 7) throw Exception("Cell array of coefficients is too small.");
  for (size_t i = 0; i < 7; i++) {
    const prost_value* cv = cell_arr->cells[i];
    if (!cv || cv->kind != PROST_VALUE_MATRIX) throw Exception("Argument has to be passed as a vector of type single or double.");
    if (cv->cols != 1 && cv->rows != 1) throw Exception("Vector has to be Nx1 or 1xN.");
    if (cv->rows == 0 || cv->cols == 0) throw Exception("Empty vector passed.");
    // narrowed to T straight from the value's storage on several host threads (a per-pixel coefficient of a 2048 x 2048 x 64
    // volume is 2 GB of doubles: no intermediate copy)
    coeffs[i].resize(cv->data.size());
    const double* src = cv->data.data();
    T* dst = coeffs[i].data();
    ParallelFor(coeffs[i].size(), [&](size_t b, size_t e) { for (size_t k = b; k < e; k++) dst[k] = (T)src[k]; });
    if (coeffs[i].size() != 1 && coeffs[i].size() != count) throw Exception("Size of coefficients should be either 1 or count.");
  }
}

template <typename T>
static Prox<T>* make_elem(int op, int fn, size_t idx, size_t size, bool diagsteps, const prost_value* data) {
  const size_t count = (size_t)GetScalarFromCell(data, 0), dim = (size_t)GetScalarFromCell(data, 1);
  const bool interleaved = GetScalarFromCell(data, 2) > 0.;
  std::array<std::vector<T>, 7> coeffs;
  get_coefficients<T>(coeffs, cell_at(data, 3), op == PROST_OP_1D ? size : count);     // factory.cpp:326-327 / :341-342
  return new ProxElemDispatch<T>(op, fn, idx, count, dim, interleaved, diagsteps, coeffs);
}

template <typename T>
std::map<std::string, typename Factory<T>::ProxFactory>& Factory<T>::prox_reg() {
  static std::map<std::string, ProxFactory> reg;
  static bool init = false;
  if (!init) {
    init = true;
    for (int fn = 0; fn < PROST_FN_COUNT; fn++) {
      reg[std::string("elem_operation:1d:") + kFunctionNames[fn]] = [fn](size_t idx, size_t size, bool ds, const prost_value* d) { return make_elem<T>(PROST_OP_1D, fn, idx, size, ds, d); };
      reg[std::string("elem_operation:norm2:") + kFunctionNames[fn]] = [fn](size_t idx, size_t size, bool ds, const prost_value* d) { return make_elem<T>(PROST_OP_NORM2, fn, idx, size, ds, d); };
    }
    reg["moreau"] = [](size_t, size_t, bool, const prost_value* d) -> Prox<T>* { return new ProxMoreau<T>(Factory<T>::CreateProx(cell_at(d, 0))); };
    reg["zero"] = [](size_t idx, size_t size, bool, const prost_value*) -> Prox<T>* { return new ProxZero<T>(idx, size); };
    reg["elem_operation:ind_sum"] = [](size_t idx, size_t, bool ds, const prost_value* d) -> Prox<T>* {
      return new ProxElemIndSum<T>(idx, (size_t)GetScalarFromCell(d, 0), (size_t)GetScalarFromCell(d, 1), GetScalarFromCell(d, 2) > 0., ds);
    };
    reg["elem_operation:ind_simplex"] = [](size_t idx, size_t, bool ds, const prost_value* d) -> Prox<T>* {
      return new ProxElemIndSimplex<T>(idx, (size_t)GetScalarFromCell(d, 0), (size_t)GetScalarFromCell(d, 1), GetScalarFromCell(d, 2) > 0., ds);
    };
    reg["transform"] = [](size_t, size_t size, bool, const prost_value* d) -> Prox<T>* {          // factory.cpp:301-310
      std::array<std::vector<T>, 5> co;
      for (size_t i = 0; i < 5; i++) {
        std::vector<double> v = GetVector(cell_at(d, i));
        co[i] = std::vector<T>(v.begin(), v.end());
        if (co[i].size() != 1 && co[i].size() != size) throw Exception("Size of coefficients should be either 1 or count.");
      }
      return new ProxTransform<T>(Factory<T>::CreateProx(cell_at(d, 5)), co[0], co[1], co[2], co[3], co[4]);
    };
    reg["permute"] = [](size_t, size_t, bool, const prost_value* d) -> Prox<T>* {                  // factory.cpp:293-299
      std::vector<double> v = GetVector(cell_at(d, 1));
      return new ProxPermute<T>(Factory<T>::CreateProx(cell_at(d, 0)), std::vector<int32_t>(v.begin(), v.end()));
    };
    reg["ind_halfspace"] = [](size_t idx, size_t, bool ds, const prost_value* d) -> Prox<T>* {     // factory.cpp:484-496
      const size_t count = (size_t)GetScalarFromCell(d, 0), dim = (size_t)GetScalarFromCell(d, 1);
      const bool interleaved = GetScalarFromCell(d, 2) > 0.;
      const prost_value* co = cell_at(d, 3);
      std::vector<double> a = GetVector(cell_at(co, 0)), b = GetVector(cell_at(co, 1));
      return new ProxIndHalfspace<T>(idx, count, dim, interleaved, ds, std::vector<T>(a.begin(), a.end()), std::vector<T>(b.begin(), b.end()));
    };
    reg["ind_soc"] = [](size_t idx, size_t, bool ds, const prost_value* d) -> Prox<T>* {           // factory.cpp:446-456
      return new ProxIndSOC<T>(idx, (size_t)GetScalarFromCell(d, 0), (size_t)GetScalarFromCell(d, 1), GetScalarFromCell(d, 2) > 0., ds,
                               (T)GetScalarFromCell(d, 3));
    };
    reg["ind_sum"] = [](size_t idx, size_t size, bool, const prost_value* d) -> Prox<T>* {         // factory.cpp:458-481
      if (!d || d->kind != PROST_VALUE_CELL) throw Exception("Cell array expected.");
      const size_t dim = (size_t)GetScalarFromCell(d, 0);
      std::vector<double> v = GetVector(cell_at(d, 1));
      std::vector<uint64_t> inds(v.begin(), v.end());
      const T sum = (T)GetScalarFromCell(d, 2);
      const size_t count = dim ? inds.size() / dim : 0;     // "hacky" in the reference too (:466)
      if (d->cells.size() == 3) return new ProxIndSum<T>(idx, size, count, dim, inds, sum);
      if (d->cells.size() == 6) {
        const size_t dim2 = (size_t)GetScalarFromCell(d, 3);
        std::vector<double> v2 = GetVector(cell_at(d, 4));
        std::vector<uint64_t> inds2(v2.begin(), v2.end());
        return new ProxIndSum<T>(idx, size, count, dim, inds, sum, dim2 ? inds2.size() / dim2 : 0, dim2, inds2, (T)GetScalarFromCell(d, 5));
      }
      throw Exception("ind_sum: 3 or 6 data entries expected.");
    };
    reg["ind_epi_quad"] = [](size_t idx, size_t, bool ds, const prost_value* d) -> Prox<T>* {
      const size_t count = (size_t)GetScalarFromCell(d, 0), dim = (size_t)GetScalarFromCell(d, 1);
      const bool interleaved = GetScalarFromCell(d, 2) > 0.;
      const prost_value* co = cell_at(d, 3);
      std::vector<double> a = GetVector(cell_at(co, 0)), b = GetVector(cell_at(co, 1)), c = GetVector(cell_at(co, 2));
      return new ProxIndEpiQuad<T>(idx, count, dim, interleaved, ds, std::vector<T>(a.begin(), a.end()), std::vector<T>(b.begin(), b.end()), std::vector<T>(c.begin(), c.end()));
    };
  }
  return reg;
}

template <typename T>
std::map<std::string, typename Factory<T>::BlockFactory>& Factory<T>::block_reg() {
  static std::map<std::string, BlockFactory> reg;
  static bool init = false;
  if (!init) {
    init = true;
    reg["gradient2d"] = [](size_t row, size_t col, const prost_value* d) -> Block<T>* {
      return new BlockGradient2D<T>(row, col, (size_t)GetScalarFromCell(d, 0), (size_t)GetScalarFromCell(d, 1), (size_t)GetScalarFromCell(d, 2), GetScalarFromCell(d, 3) > 0.);
    };
    reg["gradient3d"] = [](size_t row, size_t col, const prost_value* d) -> Block<T>* {
      return new BlockGradient3D<T>(row, col, (size_t)GetScalarFromCell(d, 0), (size_t)GetScalarFromCell(d, 1), (size_t)GetScalarFromCell(d, 2), GetScalarFromCell(d, 3) > 0.);
    };
    reg["zero"] = [](size_t row, size_t col, const prost_value* d) -> Block<T>* {
      return new BlockZero<T>(row, col, (size_t)GetScalarFromCell(d, 0), (size_t)GetScalarFromCell(d, 1));
    };
    reg["diags"] = [](size_t row, size_t col, const prost_value* d) -> Block<T>* {            // factory.cpp:573-588
      const size_t nrows = (size_t)GetScalarFromCell(d, 0), ncols = (size_t)GetScalarFromCell(d, 1);
      std::vector<double> f = GetVector(cell_at(d, 2)), o = GetVector(cell_at(d, 3));
      if (f.size() != o.size()) throw Exception("Mismatch of size(factors) and size(offsets).");
      std::vector<int64_t> ofs(o.size());
      for (size_t i = 0; i < o.size(); i++) ofs[i] = (int64_t)o[i];
      return new BlockDiags<T>(row, col, nrows, ncols, f.size(), ofs, std::vector<T>(f.begin(), f.end()));
    };
    reg["sparse"] = [](size_t row, size_t col, const prost_value* d) -> Block<T>* {           // factory.cpp:625-655
      const prost_value* pm = cell_at(d, 0);
      if (pm->kind != PROST_VALUE_SPARSE) throw Exception("Matrix must be sparse!");
      const int nrows = (int)pm->rows, ncols = (int)pm->cols, nnz = (int)pm->jc[ncols];
      // double -> T and int64 -> int32 (narrowing as the reference) on all host cores, the arrays handed over without another copy
      std::vector<T> val((size_t)nnz);
      std::vector<int32_t> ptr(pm->jc.begin(), pm->jc.end()), ind((size_t)nnz);
      const double* src_v = pm->data.data(); const int64_t* src_i = pm->ir.data();
      ParallelFor((size_t)nnz, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; k++) { val[k] = (T)src_v[k]; ind[k] = (int32_t)src_i[k]; } });
      return BlockSparse<T>::CreateFromCSC(row, col, nrows, ncols, nnz, std::move(val), std::move(ptr), std::move(ind));
    };
    for (int id_first = 0; id_first < 2; id_first++)                                          // factory.cpp:657-755
      reg[id_first ? "id_kron_sparse" : "sparse_kron_id"] = [id_first](size_t row, size_t col, const prost_value* d) -> Block<T>* {
        const prost_value* pm = cell_at(d, 0);
        if (pm->kind != PROST_VALUE_SPARSE) throw Exception("Matrix must be sparse!");
        const int nrows = (int)pm->rows, ncols = (int)pm->cols, nnz = (int)pm->jc[ncols];
        const size_t diaglength = (size_t)GetScalarFromCell(d, 1);
        std::vector<T> val(pm->data.begin(), pm->data.begin() + nnz);
        std::vector<int32_t> ptr(pm->jc.begin(), pm->jc.end()), ind(pm->ir.begin(), pm->ir.begin() + nnz);
        return BlockKronSparse<T>::CreateFromCSC(id_first != 0, row, col, diaglength, nrows, ncols, nnz, val, ptr, ind);
      };
  }
  return reg;
}

template <typename T>
std::map<std::string, typename Factory<T>::BackendFactory>& Factory<T>::backend_reg() {
  static std::map<std::string, BackendFactory> reg;
  static bool init = false;
  if (!init) {
    init = true;
    reg["pdhg"] = [](const prost_value* d) -> Backend<T>* {                                   // factory.cpp:766-797
      typename BackendPDHG<T>::Options o;
      o.tau0 = GetScalarFromField(d, "tau0"); o.sigma0 = GetScalarFromField(d, "sigma0");
      o.residual_iter = (int)GetScalarFromField(d, "residual_iter");
      o.scale_steps_operator = GetScalarFromField(d, "scale_steps_operator") > 0.;
      o.alg2_gamma = (T)GetScalarFromField(d, "alg2_gamma");
      o.arg_alpha0 = (T)GetScalarFromField(d, "arg_alpha0"); o.arg_nu = (T)GetScalarFromField(d, "arg_nu");
      o.arg_delta = (T)GetScalarFromField(d, "arg_delta"); o.arb_delta = (T)GetScalarFromField(d, "arb_delta");
      o.arb_tau = (T)GetScalarFromField(d, "arb_tau");
      const std::string sv = GetString(prost_value_field(d, "stepsize"));
      if (sv == "alg1") o.stepsize_variant = BackendPDHG<T>::kPDHGStepsAlg1;
      else if (sv == "alg2") o.stepsize_variant = BackendPDHG<T>::kPDHGStepsAlg2;
      else if (sv == "goldstein") o.stepsize_variant = BackendPDHG<T>::kPDHGStepsResidualGoldstein;
      else if (sv == "boyd") o.stepsize_variant = BackendPDHG<T>::kPDHGStepsResidualBoyd;
      else throw Exception("Couldn't recognize step-size variant. Valid options are {alg1,alg2,goldstein,boyd}.");
      if (prost_value_field(d, "allow_fused")) o.allow_fused = GetScalarFromField(d, "allow_fused") > 0.;
      if (prost_value_field(d, "allow_single_kernel")) o.allow_single_kernel = GetScalarFromField(d, "allow_single_kernel") > 0.;
      if (prost_value_field(d, "allow_pair_kernel")) o.allow_pair_kernel = GetScalarFromField(d, "allow_pair_kernel") > 0.;
      if (prost_value_field(d, "allow_speculation")) o.allow_speculation = GetScalarFromField(d, "allow_speculation") > 0.;
      if (prost_value_field(d, "allow_arg_fusion")) o.allow_arg_fusion = GetScalarFromField(d, "allow_arg_fusion") > 0.;
      if (prost_value_field(d, "allow_op_fusion")) o.allow_op_fusion = (int)GetScalarFromField(d, "allow_op_fusion");
      if (prost_value_field(d, "residual_sums_in_prox")) o.residual_sums_in_prox = (int)GetScalarFromField(d, "residual_sums_in_prox");
      if (prost_value_field(d, "allow_device_rules")) o.allow_device_rules = GetScalarFromField(d, "allow_device_rules") > 0.;
      if (prost_value_field(d, "group_max")) o.group_max = (int)GetScalarFromField(d, "group_max");
      if (prost_value_field(d, "arithmetic")) {            // MI355X addition: 'exact' (default) | 'fmad' (tolerance class, BackendPDHG::Options::arithmetic)
        const std::string ar = GetString(prost_value_field(d, "arithmetic"));
        if (ar == "exact") o.arithmetic = PROST_HIP_ARITH_EXACT;
        else if (ar == "fmad") o.arithmetic = PROST_HIP_ARITH_FMAD;
        else throw Exception("Couldn't recognize arithmetic class. Valid options are {exact,fmad}.");
      }
      return new BackendPDHG<T>(o);
    };
    reg["admm"] = [](const prost_value* d) -> Backend<T>* {                                   // factory.cpp:799-818
      typename BackendADMM<T>::Options o;
      o.rho0 = GetScalarFromField(d, "rho0"); o.residual_iter = (int)GetScalarFromField(d, "residual_iter");
      o.arb_delta = (T)GetScalarFromField(d, "arb_delta"); o.arb_gamma = (T)GetScalarFromField(d, "arb_gamma");
      o.arb_tau = (T)GetScalarFromField(d, "arb_tau"); o.alpha = GetScalarFromField(d, "alpha");
      o.cg_max_iter = (int)GetScalarFromField(d, "cg_max_iter"); o.cg_tol_pow = GetScalarFromField(d, "cg_tol_pow");
      o.cg_tol_min = GetScalarFromField(d, "cg_tol_min"); o.cg_tol_max = GetScalarFromField(d, "cg_tol_max");
      if (prost_value_field(d, "device_cg")) o.device_cg = GetScalarFromField(d, "device_cg") > 0.;
      if (prost_value_field(d, "cg_graph")) o.cg_graph = GetScalarFromField(d, "cg_graph") > 0.;
      if (prost_value_field(d, "fused_rounds")) o.fused_rounds = GetScalarFromField(d, "fused_rounds") > 0.;
      if (prost_value_field(d, "pixel_rounds")) o.pixel_rounds = GetScalarFromField(d, "pixel_rounds") > 0.;
      return new BackendADMM<T>(o);
    };
  }
  return reg;
}

template <class Reg>
static std::string names_of(const Reg& reg) {
  std::ostringstream ss;
  bool first = true;
  for (auto& e : reg) { if (!first) ss << ", "; ss << e.first; first = false; }
  return ss.str();
}

template <typename T>
shared_ptr<Prox<T>> Factory<T>::CreateProx(const prost_value* pm) {
  if (!pm || pm->kind != PROST_VALUE_CELL || pm->cells.size() != 5) {
    std::stringstream ss; ss << "Invalid prox description. Dim = " << (pm ? pm->cells.size() : 0) << " (should be 5).";
    throw Exception(ss.str());
  }
  const std::string name = GetString(cell_at(pm, 0));
  const size_t idx = (size_t)GetScalarFromCell(pm, 1), size = (size_t)GetScalarFromCell(pm, 2);
  const bool diagsteps = GetScalarFromCell(pm, 3) > 0.;
  const prost_value* data = pm->cells[4];
  Prox<T>* prox = nullptr;
  auto it = prox_reg().find(name);
  if (it != prox_reg().end()) {
    try { prox = it->second(idx, size, diagsteps, data); }
    catch (Exception& e) { throw Exception("Creating prox with ID '" + name + "' failed. Reason: " + e.what()); }
  }
  if (!prox) throw Exception("Creating prox with ID '" + name + "' failed. Reason: Name not registered in ProxFactory. Available prox are: { " + names_of(prox_reg()) + " }.\n");
  return shared_ptr<Prox<T>>(prox);
}

template <typename T>
shared_ptr<Block<T>> Factory<T>::CreateBlock(const prost_value* pm) {
  if (!pm || pm->kind != PROST_VALUE_CELL || pm->cells.size() != 4) throw Exception("Invalid block description. Dim != 4.");
  const std::string name = GetString(cell_at(pm, 0));
  const size_t row = (size_t)GetScalarFromCell(pm, 1), col = (size_t)GetScalarFromCell(pm, 2);
  Block<T>* block = nullptr;
  auto it = block_reg().find(name);
  if (it != block_reg().end()) {
    try { block = it->second(row, col, pm->cells[3]); }
    catch (Exception& e) { throw Exception("Creating block with ID '" + name + "' failed. Reason: " + e.what()); }
  }
  if (!block) throw Exception("Creating block with ID '" + name + "' failed. Reason: Name not registered in BlockFactory. Available blocks are: { " + names_of(block_reg()) + " }.\n");
  return shared_ptr<Block<T>>(block);
}

template <typename T>
shared_ptr<Backend<T>> Factory<T>::CreateBackend(const prost_value* pm) {
  std::string name = GetString(cell_at(pm, 0));
  std::transform(name.begin(), name.end(), name.begin(), ::tolower);
  Backend<T>* backend = nullptr;
  auto it = backend_reg().find(name);
  if (it != backend_reg().end()) {
    try { backend = it->second(cell_at(pm, 1)); }
    catch (Exception& e) { throw Exception("Creating backend with ID '" + name + "' failed. Reason: " + e.what()); }
  }
  if (!backend) throw Exception("Creating backend with ID '" + name + "' failed. Reason: Name not registered in BackendFactory. Available backends are: { " + names_of(backend_reg()) + " }.\n");
  return shared_ptr<Backend<T>>(backend);
}

template <typename T>
shared_ptr<Problem<T>> Factory<T>::CreateProblem(const prost_value* pm, size_t nrows, size_t ncols) {
  shared_ptr<Problem<T>> prob(new Problem<T>);
  for (auto* b : cell_list(prost_value_field(pm, "linop"))) prob->AddBlock(CreateBlock(b));
  for (auto* p : cell_list(prost_value_field(pm, "prox_g"))) prob->AddProx_g(CreateProx(p));
  for (auto* p : cell_list(prost_value_field(pm, "prox_f"))) prob->AddProx_f(CreateProx(p));
  for (auto* p : cell_list(prost_value_field(pm, "prox_gstar"))) prob->AddProx_gstar(CreateProx(p));
  for (auto* p : cell_list(prost_value_field(pm, "prox_fstar"))) prob->AddProx_fstar(CreateProx(p));
  const std::string scaling = GetString(prost_value_field(pm, "scaling"));
  if (scaling == "alpha") prob->SetScalingAlpha((T)GetScalarFromField(pm, "scaling_alpha"));
  else if (scaling == "identity") prob->SetScalingIdentity();
  else if (scaling == "custom") {
    std::vector<double> l = GetVector(prost_value_field(pm, "scaling_left")), r = GetVector(prost_value_field(pm, "scaling_right"));
    prob->SetScalingCustom(std::vector<T>(l.begin(), l.end()), std::vector<T>(r.begin(), r.end()));
  } else throw Exception("Problem scaling variant not recognized. Options are {'alpha', 'identity', 'custom'}.");
  prob->SetDimensions(nrows, ncols);
  return prob;
}

template <typename T>
typename Solver<T>::Options Factory<T>::CreateSolverOptions(const prost_value* pm) {
  typename Solver<T>::Options o;
  o.tol_rel_primal = (T)GetScalarFromField(pm, "tol_rel_primal"); o.tol_rel_dual = (T)GetScalarFromField(pm, "tol_rel_dual");
  o.tol_abs_primal = (T)GetScalarFromField(pm, "tol_abs_primal"); o.tol_abs_dual = (T)GetScalarFromField(pm, "tol_abs_dual");
  o.max_iters = (int)GetScalarFromField(pm, "max_iters"); o.num_cback_calls = (int)GetScalarFromField(pm, "num_cback_calls");
  o.verbose = GetScalarFromField(pm, "verbose") > 0.; o.solve_dual_problem = GetScalarFromField(pm, "solve_dual") > 0.;
  const prost_value* x0 = prost_value_field(pm, "x0");
  const prost_value* y0 = prost_value_field(pm, "y0");
  if (x0 && x0->kind == PROST_VALUE_MATRIX && x0->rows > 0) { auto v = GetVector(x0); o.x0 = std::vector<T>(v.begin(), v.end()); }
  if (y0 && y0->kind == PROST_VALUE_MATRIX && y0->rows > 0) { auto v = GetVector(y0); o.y0 = std::vector<T>(v.begin(), v.end()); }
  return o;
}

template struct Factory<float>;
template struct Factory<double>;

}  // namespace prost

// ------------------------------------------------------------------------------------------
// command table
// ------------------------------------------------------------------------------------------
namespace {

int g_device = 0;
bool g_single = false;       // reference default: typedef double real (matlab/+prost/private/config.hpp:7)
void* g_comm = nullptr;
int g_comm_world = 1;

#define CMD_ARGS int nlhs, prost_value** plhs, int nrhs, const prost_value* const* prhs

void select_device() {
  int n = 0;
  if (prost_hip_device_count(&n) != 0 || n < 1) throw Exception("Invalid HIP device: no MI355X visible (the prost hot path has no CPU fallback).");
  if (g_device >= n) throw Exception("Invalid HIP device.");
  CheckHip(prost_hip_set_device(g_device), "set_device");
}

prost_value* vec_value(const std::vector<double>& v) { return prost_value_matrix(v.data(), v.size(), 1); }
// widens straight into the value's storage (one pass; these are the 10^7..10^8-element result vectors)
template <typename T> prost_value* vec_value_t(const std::vector<T>& v) {
  prost_value* out = new prost_value; out->kind = PROST_VALUE_MATRIX; out->rows = v.size(); out->cols = 1;
  out->data.resize(v.size());           // no zero fill (default_init_allocator)
  double* dst = out->data.data();
  const T* src = v.data();
  ParallelFor(v.size(), [&](size_t b, size_t e) { for (size_t i = b; i < e; i++) dst[i] = (double)src[i]; });
  return out;
}

template <typename T>
struct SolverHandle {
  shared_ptr<Problem<T>> problem;
  shared_ptr<Backend<T>> backend;
  shared_ptr<Solver<T>> solver;
};
struct AnyHandle { bool single; shared_ptr<void> h; };
std::map<int, AnyHandle> g_handles;
int g_next_handle = 1;

template <typename T>
shared_ptr<SolverHandle<T>> build_solver(const prost_value* problem, size_t nrows, size_t ncols, const prost_value* backend,
                                         const prost_value* opts, bool with_callbacks, const prost_value* owned = nullptr) {
  auto h = std::make_shared<SolverHandle<T>>();
  { StageTimer t("factory: CreateProblem"); h->problem = Factory<T>::CreateProblem(problem, nrows, ncols); }
  h->backend = Factory<T>::CreateBackend(backend);
  typename Solver<T>::Options o = Factory<T>::CreateSolverOptions(opts);
  if (o.verbose) {
    std::cout << "prost v" << get_version() << std::endl;
    char name[256]; int cus = 0; size_t mem = 0;
    if (prost_hip_device_info(g_device, name, sizeof(name), &cus, &mem) == 0)
      std::printf("Running on device number %d: %s (%.1f GB, %d CUs), float precision: %d bit.\n", g_device, name,
                  (double)mem / (1024. * 1024 * 1024), cus, (int)sizeof(T) * 8);
  }
  h->solver = std::make_shared<Solver<T>>(h->problem, h->backend);
  h->solver->SetOptions(o);
  if (with_callbacks) {
    const prost_value* cb = prost_value_field(opts, "interm_cb");
    if (cb && cb->kind == PROST_VALUE_CALLBACK && cb->cb) {
      prost_interm_cb fn = cb->cb; void* user = cb->cb_user;
      h->solver->SetIntermCallback([fn, user](int it, const std::vector<T>& x, const std::vector<T>& y) {
        std::vector<double> dx(x.begin(), x.end()), dy(y.begin(), y.end());
        return fn(user, it, dx.data(), dx.size(), dy.data(), dy.size()) != 0;
      });
    }
    // (options.m's default dummy_cb prints a newline and returns false: exactly what Solver::Solve does by itself when no
    // callback is installed, so none is -- and the final read-out can stream the result, see solve_problem_t)
    // only when the front end registered one (the MEX gateway polls Ctrl-C, prost.cpp:58-66): it is asked once per launch
    if (g_stop_cb) h->solver->SetStoppingCallback([]() { return g_stop_cb ? g_stop_cb(g_stop_user) != 0 : false; });
  }
  // column-sharded images: only the owned columns of this slab count (residual sums, global sizes)
  double own_frac = 1.0;
  if (owned && owned->kind == PROST_VALUE_MATRIX && owned->data.size() >= 3) {
    auto* pd = dynamic_cast<BackendPDHG<T>*>(h->backend.get());
    if (!pd) throw Exception("Owned columns are only supported by the pdhg backend.");
    const size_t x0 = (size_t)owned->data[0], x1 = (size_t)owned->data[1], nx = (size_t)owned->data[2];
    if (x1 <= x0 || x1 > nx) throw Exception("Owned column range must satisfy x0 < x1 <= nx.");
    pd->SetOwnedColumns(x0, x1);
    own_frac = (double)(x1 - x0) / (double)nx;
  }
  if (g_comm) {
    // global sizes for eps_primal / eps_dual: sum over ranks
    double* d = nullptr; double hbuf[2] = {(double)nrows * own_frac, (double)ncols * own_frac};
    CheckHip(prost_hip_malloc((void**)&d, 2 * sizeof(double)), "malloc");
    CheckHip(prost_hip_memcpy_h2d(d, hbuf, sizeof(hbuf), nullptr), "h2d");
    CheckHip(prost_hip_allreduce_sum_f64(g_comm, d, 2, nullptr), "allreduce");
    CheckHip(prost_hip_memcpy_d2h(hbuf, d, sizeof(hbuf), nullptr), "d2h");
    CheckHip(prost_hip_stream_synchronize(nullptr), "sync");
    prost_hip_free(d);
    h->backend->SetCommunicator(g_comm, (size_t)hbuf[0], (size_t)hbuf[1]);
  }
  { StageTimer t("Solver::Initialize (total)"); h->solver->Initialize(); }
  if (own_frac != 1.0) {
    auto* pd = dynamic_cast<BackendPDHG<T>*>(h->backend.get());
    if (!pd->sharded_path()) throw Exception("Column sharding needs a one-kernel gradient2d path (one gradient2d block with L <= 4 channels).");
  }
  return h;
}

template <typename T>
void solve_problem_t(CMD_ARGS) {
  select_device();
  const size_t nrows = (size_t)prhs[1]->data[0], ncols = (size_t)prhs[2]->data[0];
  // PROST_TRACE_SOLVE=1: wall time of the stages of a solve on stderr (tools/time_to_solution.py)
  const bool trace = std::getenv("PROST_TRACE_SOLVE") != nullptr;
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const auto t0 = now();
  auto h = build_solver<T>(prhs[0], nrows, ncols, prhs[3], prhs[4], true);
  const auto t1 = now();
  // The result goes from the device straight into the double-precision values the caller gets (pinned staging, widened by
  // the copying threads) instead of through four host vectors of T and a widening pass -- unless an intermediate-solution
  // callback needs those vectors anyway, or the backend keeps no device-resident solution.
  prost_value* streamed[4] = {nullptr, nullptr, nullptr, nullptr};        // x, z, y, w of the solved (possibly dualised) problem
  h->solver->SetFinalReadout([&]() {
    const T* dev[4];
    const auto r0 = now();
    if (!h->backend->current_solution_device(dev[0], dev[1], dev[2], dev[3])) return false;
    const auto r1 = now();
    const size_t n = h->problem->ncols(), m = h->problem->nrows();
    const size_t len[4] = {n, m, m, n};
    for (int k = 0; k < 4; k++) {
      prost_value* v = new prost_value; v->kind = PROST_VALUE_MATRIX; v->rows = len[k]; v->cols = 1;
      v->data.resize(len[k]);             // no zero fill (default_init_allocator)
      streamed[k] = v;
      DownloadAs<double, T>(v->data.data(), dev[k], len[k]);
    }
    if (trace) std::fprintf(stderr, "solve_problem: read-out: constraint variables %.3f s, download + widening of x, z, y, w %.3f s\n", secs(r0, r1), secs(r1, now()));
    return true;
  });
  typename Solver<T>::ConvergenceResult r;
  try { r = h->solver->Solve(); } catch (...) { for (auto* v : streamed) if (v) prost_value_free(v); throw; }
  const auto t2 = now();
  prost_value* out = prost_value_struct();
  if (streamed[0]) {
    // under solve_dual the roles of the solution vectors are exchanged (solver.cu:216-246)
    const bool dual = h->solver->options().solve_dual_problem;
    prost_value_struct_set(out, "x", streamed[dual ? 2 : 0]);
    prost_value_struct_set(out, "y", streamed[dual ? 0 : 2]);
    prost_value_struct_set(out, "z", streamed[dual ? 3 : 1]);
    prost_value_struct_set(out, "w", streamed[dual ? 1 : 3]);
  } else {
  prost_value_struct_set(out, "x", vec_value_t(h->solver->cur_primal_sol()));
  prost_value_struct_set(out, "y", vec_value_t(h->solver->cur_dual_sol()));
  prost_value_struct_set(out, "z", vec_value_t(h->solver->cur_primal_constr_sol()));
  prost_value_struct_set(out, "w", vec_value_t(h->solver->cur_dual_constr_sol()));
  }
  if (trace) std::fprintf(stderr, "solve_problem: build + initialize %.3f s, Solve (iterations + final read-out) %.3f s, widening the result %.3f s\n", secs(t0, t1), secs(t1, t2), secs(t2, now()));
  const char* msg = r == Solver<T>::kConverged ? "Converged." : (r == Solver<T>::kStoppedMaxIters ? "Reached maximum iterations." : "Stopped by user.");
  prost_value_struct_set(out, "result", prost_value_string(msg));
  prost_value_struct_set(out, "iters", prost_value_scalar(h->solver->iterations_done()));
  prost_value_struct_set(out, "path", prost_value_string(h->backend->path().c_str()));
  prost_value_struct_set(out, "pair_launches", prost_value_scalar((double)h->backend->pair_launches()));
  h->solver->Release();
  if (nlhs >= 1) plhs[0] = out; else prost_value_free(out);
}
void cmd_solve_problem(CMD_ARGS) {
  if (nrhs != 5) throw Exception("solve_problem: five inputs (problem, nrows, ncols, backend, opts) required.");
  if (g_single) solve_problem_t<float>(nlhs, plhs, nrhs, prhs); else solve_problem_t<double>(nlhs, plhs, nrhs, prhs);
}

template <typename T>
void eval_linop_t(CMD_ARGS) {
  select_device();
  shared_ptr<LinearOperator<T>> linop(new LinearOperator<T>());
  for (auto* c : cell_list(prhs[0])) linop->AddBlock(Factory<T>::CreateBlock(c));
  const bool transpose = prhs[2]->data[0] > 0;
  if (prhs[1]->cols != 1) throw Exception("Right-hand side input to eval_linop should be a n-times-1 vector!");
  linop->Initialize();
  const size_t need = transpose ? linop->nrows() : linop->ncols();
  if (prhs[1]->data.size() < need) throw Exception("Right-hand side input to eval_linop is too short.");
  std::vector<T> rhs(prhs[1]->data.begin(), prhs[1]->data.begin() + need), res;
  const double time = transpose ? linop->EvalAdjoint(res, rhs) : linop->Eval(res, rhs);
  std::vector<double> rowsum(linop->nrows()), colsum(linop->ncols());
  for (size_t r = 0; r < linop->nrows(); r++) rowsum[r] = linop->row_sum(r, 1);
  for (size_t c = 0; c < linop->ncols(); c++) colsum[c] = linop->col_sum(c, 1);
  linop->Release();
  plhs[0] = vec_value_t(res); plhs[1] = vec_value(rowsum); plhs[2] = vec_value(colsum);
  if (nlhs >= 4) plhs[3] = prost_value_scalar(time);
}
void cmd_eval_linop(CMD_ARGS) {
  if (nrhs != 3) throw Exception("eval_lin_op: Three inputs required!");
  if (nlhs < 3) throw Exception("eval_lin_op: At least three outputs (result, rowsum, colsum) required.");
  if (g_single) eval_linop_t<float>(nlhs, plhs, nrhs, prhs); else eval_linop_t<double>(nlhs, plhs, nrhs, prhs);
}

template <typename T>
void eval_prox_t(CMD_ARGS) {
  select_device();
  const size_t n = prhs[1]->rows;
  if (prhs[1]->cols != 1) throw Exception("Input to prox should be a vector!");
  shared_ptr<Prox<T>> prox = Factory<T>::CreateProx(prhs[0]);
  prox->Initialize();
  if (prox->size() != n) {
    std::stringstream ss; ss << "Size of input argument (" << n << ") doesn't match size of prox (" << prox->size() << ")!\n";
    throw Exception(ss.str());
  }
  if (prhs[3]->data.size() < n) throw Exception("Diagonal step size vector is too short.");
  std::vector<T> arg(prhs[1]->data.begin(), prhs[1]->data.end()), tau(prhs[3]->data.begin(), prhs[3]->data.begin() + n), res;
  const double ms = prox->Eval(res, arg, tau, (T)prhs[2]->data[0]);
  prox->Release();
  plhs[0] = vec_value_t(res);
  if (nlhs >= 2) plhs[1] = prost_value_scalar(ms);
}
void cmd_eval_prox(CMD_ARGS) {
  if (nrhs < 4) throw Exception("eval_prox: At least four inputs required.");
  if (nlhs == 0) throw Exception("One output (result of prox) required.");
  if (g_single) eval_prox_t<float>(nlhs, plhs, nrhs, prhs); else eval_prox_t<double>(nlhs, plhs, nrhs, prhs);
}

void cmd_init(CMD_ARGS) { (void)nlhs; (void)plhs; (void)nrhs; (void)prhs; }
void cmd_release(CMD_ARGS) { (void)nlhs; (void)plhs; (void)nrhs; (void)prhs; g_handles.clear(); }
void cmd_list_gpus(CMD_ARGS) {
  (void)nrhs; (void)prhs;
  int n = 0; prost_hip_device_count(&n);
  for (int i = 0; i < n; i++) {
    char name[256]; int cus = 0; size_t mem = 0;
    CheckHip(prost_hip_device_info(i, name, sizeof(name), &cus, &mem), "device_info");
    std::printf("Device number %d: %s (%.1f GB, %d CUs).\n", i, name, (double)mem / (1024. * 1024 * 1024), cus);
  }
  if (nlhs >= 1) plhs[0] = prost_value_scalar(n);
}
void cmd_set_gpu(CMD_ARGS) { (void)nlhs; (void)plhs; if (nrhs < 1) throw Exception("set_gpu: device id required."); g_device = (int)prhs[0]->data[0]; }
void cmd_set_precision(CMD_ARGS) {
  (void)nlhs; (void)plhs;
  if (nrhs < 1) throw Exception("set_precision: 'single' or 'double' required.");
  const std::string s = GetString(prhs[0]);
  if (s == "single" || s == "float") g_single = true;
  else if (s == "double") g_single = false;
  else throw Exception("set_precision: 'single' or 'double' required.");
}
void cmd_get_precision(CMD_ARGS) { (void)nrhs; (void)prhs; if (nlhs >= 1) plhs[0] = prost_value_string(g_single ? "single" : "double"); }

template <typename T>
prost_value* prox_rows(const typename Problem<T>::ProxList& l) {
  std::vector<double> d;
  for (auto& p : l) { d.push_back((double)p->index()); d.push_back((double)p->size()); d.push_back(p->diagsteps() ? 1 : 0); }
  return prost_value_matrix(d.data(), 3, l.size());
}
template <typename T>
void problem_info_t(CMD_ARGS) {
  (void)nrhs;
  auto prob = Factory<T>::CreateProblem(prhs[0], (size_t)prhs[1]->data[0], (size_t)prhs[2]->data[0]);
  prob->InitializeHost();
  prost_value* out = prost_value_struct();
  prost_value_struct_set(out, "scaling_left", vec_value_t(prob->scaling_left_host()));
  prost_value_struct_set(out, "scaling_right", vec_value_t(prob->scaling_right_host()));
  prost_value_struct_set(out, "nrows", prost_value_scalar((double)prob->nrows()));
  prost_value_struct_set(out, "ncols", prost_value_scalar((double)prob->ncols()));
  prost_value_struct_set(out, "linop_nrows", prost_value_scalar((double)prob->linop()->nrows()));
  prost_value_struct_set(out, "linop_ncols", prost_value_scalar((double)prob->linop()->ncols()));
  prost_value_struct_set(out, "prox_g", prox_rows<T>(prob->prox_g()));
  prost_value_struct_set(out, "prox_f", prox_rows<T>(prob->prox_f()));
  prost_value_struct_set(out, "prox_gstar", prox_rows<T>(prob->prox_gstar()));
  prost_value_struct_set(out, "prox_fstar", prox_rows<T>(prob->prox_fstar()));
  if (nlhs >= 1) plhs[0] = out; else prost_value_free(out);
}
void cmd_problem_info(CMD_ARGS) {
  if (nrhs != 3) throw Exception("problem_info: three inputs (problem, nrows, ncols) required.");
  if (g_single) problem_info_t<float>(nlhs, plhs, nrhs, prhs); else problem_info_t<double>(nlhs, plhs, nrhs, prhs);
}

/// glibc_rand_unit(n[, skip[, piece]]) -> n x 1: (T)rand() / (T)RAND_MAX of a fresh process after `skip` draws -- the start vector of
/// Problem::normest (problem.cu:441-444) as GlibcRand::fill_unit generates it (chunk-parallel by jump-ahead); host only
void cmd_glibc_rand_unit(CMD_ARGS) {
  if (nrhs < 1) throw Exception("glibc_rand_unit: n required.");
  const size_t n = (size_t)prhs[0]->data[0], skip = nrhs >= 2 ? (size_t)prhs[1]->data[0] : 0;
  const size_t piece = nrhs >= 3 && prhs[2]->data[0] > 0 ? (size_t)prhs[2]->data[0] : n;     // consecutive fill_unit calls of this length
  GlibcRand rng(1);
  for (size_t i = 0; i < skip; i++) rng.next();
  prost_value* out = prost_value_matrix(nullptr, n, 1);
  if (g_single) {
    std::vector<float> v(n);
    for (size_t b = 0; b < n; b += piece) rng.fill_unit(v.data() + b, std::min(piece, n - b));
    for (size_t i = 0; i < n; i++) out->data[i] = v[i];
  } else {
    for (size_t b = 0; b < n; b += piece) rng.fill_unit(out->data.data() + b, std::min(piece, n - b));
  }
  if (nlhs >= 1) plhs[0] = out; else prost_value_free(out);
}
void cmd_solver_create(CMD_ARGS) {
  if (nrhs != 5 && nrhs != 6) throw Exception("solver_create: five inputs (problem, nrows, ncols, backend, opts) [+ owned columns {x0, x1, nx}] required.");
  select_device();
  const size_t nrows = (size_t)prhs[1]->data[0], ncols = (size_t)prhs[2]->data[0];
  const prost_value* owned = nrhs == 6 ? prhs[5] : nullptr;
  AnyHandle a; a.single = g_single;
  if (g_single) a.h = build_solver<float>(prhs[0], nrows, ncols, prhs[3], prhs[4], false, owned);
  else a.h = build_solver<double>(prhs[0], nrows, ncols, prhs[3], prhs[4], false, owned);
  const int id = g_next_handle++;
  g_handles[id] = a;
  if (nlhs >= 1) plhs[0] = prost_value_scalar(id);
}
AnyHandle& handle_of(const prost_value* v) {
  auto it = g_handles.find((int)v->data[0]);
  if (it == g_handles.end()) throw Exception("Unknown solver handle.");
  return it->second;
}
template <typename T>
static prost_value* kernel_times_value(Backend<T>& backend) {
  std::vector<typename Backend<T>::KernelTime> kt;
  backend.KernelTimes(kt);
  // one cell per kernel kind sampled: {name, avg_ms, sampled launches, iterations per launch, all launches, chunk columns}
  prost_value* ks = prost_value_cell(kt.size());
  for (size_t i = 0; i < kt.size(); i++) {
    prost_value* e = prost_value_cell(6);
    prost_value_cell_set(e, 0, prost_value_string(kt[i].name.c_str()));
    prost_value_cell_set(e, 1, prost_value_scalar(kt[i].avg_ms));
    prost_value_cell_set(e, 2, prost_value_scalar((double)kt[i].sampled));
    prost_value_cell_set(e, 3, prost_value_scalar((double)kt[i].iterations_per_launch));
    prost_value_cell_set(e, 4, prost_value_scalar((double)kt[i].launches));
    prost_value_cell_set(e, 5, prost_value_scalar((double)kt[i].chunk_cols));
    prost_value_cell_set(ks, i, e);
  }
  return ks;
}
template <typename T>
void solver_iterate_t(SolverHandle<T>& h, int iters, bool time_kernels, int sample_every, bool checked, bool defer_times, int nlhs, prost_value** plhs) {
  h.backend->EnableKernelTiming(time_kernels, sample_every);
  CheckHip(prost_hip_stream_synchronize(CurrentStream()), "sync");
  const auto t0 = std::chrono::steady_clock::now();
  bool converged = false;
  if (checked) converged = h.solver->IterateChecked(iters); else h.solver->Iterate(iters);
  CheckHip(prost_hip_stream_synchronize(CurrentStream()), "sync");
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  CheckHip(prost_hip_check_last_error(), "solver_iterate");
  h.backend->EnableKernelTiming(false);
  prost_value* out = prost_value_struct();
  prost_value_struct_set(out, "ms", prost_value_scalar(ms));
  prost_value_struct_set(out, "converged", prost_value_scalar(converged ? 1 : 0));
  // defer_times: the event pairs stay recorded and are evaluated by solver_kernel_times (outside a caller's timed region)
  prost_value_struct_set(out, "kernels", defer_times ? prost_value_cell(0) : kernel_times_value<T>(*h.backend));
  if (nlhs >= 1) plhs[0] = out; else prost_value_free(out);
}
void cmd_solver_iterate(CMD_ARGS) {
  if (nrhs < 2) throw Exception("solver_iterate: (handle, iters) required.");
  AnyHandle& a = handle_of(prhs[0]);
  const int iters = (int)prhs[1]->data[0];
  const bool tk = nrhs >= 3 && prhs[2]->data[0] > 0;
  const int every = nrhs >= 4 ? (int)prhs[3]->data[0] : 8;
  const bool checked = nrhs >= 5 && prhs[4]->data[0] > 0;
  const bool defer = nrhs >= 6 && prhs[5]->data[0] > 0;
  if (a.single) solver_iterate_t(*std::static_pointer_cast<SolverHandle<float>>(a.h), iters, tk, every, checked, defer, nlhs, plhs);
  else solver_iterate_t(*std::static_pointer_cast<SolverHandle<double>>(a.h), iters, tk, every, checked, defer, nlhs, plhs);
}
void cmd_solver_kernel_times(CMD_ARGS) {
  if (nrhs < 1) throw Exception("solver_kernel_times: handle required.");
  AnyHandle& a = handle_of(prhs[0]);
  prost_value* out = a.single ? kernel_times_value<float>(*std::static_pointer_cast<SolverHandle<float>>(a.h)->backend)
                              : kernel_times_value<double>(*std::static_pointer_cast<SolverHandle<double>>(a.h)->backend);
  if (nlhs >= 1) plhs[0] = out; else prost_value_free(out);
}
template <typename T>
void solver_state_t(SolverHandle<T>& h, bool vectors, int nlhs, prost_value** plhs) {
  if (vectors) { StageTimer t("FetchSolution"); h.solver->FetchSolution(); }
  StageTimer t2("solver_state: value tree");
  prost_value* out = prost_value_struct();
  if (vectors) {
    prost_value_struct_set(out, "x", vec_value_t(h.solver->cur_primal_sol()));
    prost_value_struct_set(out, "y", vec_value_t(h.solver->cur_dual_sol()));
    prost_value_struct_set(out, "z", vec_value_t(h.solver->cur_primal_constr_sol()));
    prost_value_struct_set(out, "w", vec_value_t(h.solver->cur_dual_constr_sol()));
  }
  double tau = 0, sigma = 0, theta = 0, rho = 0, it = 0, cg_its = 0, spec_l = 0, spec_a = 0, dev_batches = 0, op_fused = 0, res_prox = 0, arith = 0, gmax = 0;
  if (auto* p = dynamic_cast<BackendPDHG<T>*>(h.backend.get())) {
    arith = (double)p->arithmetic(); gmax = (double)p->group_max();
    tau = p->tau(); sigma = p->sigma(); theta = p->theta(); it = (double)p->iteration();
    spec_l = (double)p->speculative_launches(); spec_a = (double)p->speculative_adopted();
    dev_batches = (double)p->device_rule_batches();
    op_fused = p->operator_in_prox_kernels() ? 1 : 0;
    res_prox = p->residual_sums_in_prox_launches() ? 1 : 0;
  }
  if (auto* a = dynamic_cast<BackendADMM<T>*>(h.backend.get())) { rho = a->rho(); it = (double)a->iteration(); cg_its = a->last_cg_iterations(); }
  const char* names[] = {"tau", "sigma", "theta", "rho", "iteration", "primal_res", "dual_res", "primal_var_norm", "dual_var_norm", "eps_primal", "eps_dual"};
  const double vals[] = {tau, sigma, theta, rho, it, (double)h.backend->primal_residual(), (double)h.backend->dual_residual(),
                         (double)h.backend->primal_var_norm(), (double)h.backend->dual_var_norm(), (double)h.backend->eps_primal(), (double)h.backend->eps_dual()};
  for (int i = 0; i < 11; i++) prost_value_struct_set(out, names[i], prost_value_scalar(vals[i]));
  prost_value_struct_set(out, "cg_iterations", prost_value_scalar(cg_its));
  prost_value_struct_set(out, "path", prost_value_string(h.backend->path().c_str()));
  prost_value_struct_set(out, "pair_launches", prost_value_scalar((double)h.backend->pair_launches()));
  prost_value_struct_set(out, "speculative_launches", prost_value_scalar(spec_l));
  prost_value_struct_set(out, "speculative_adopted", prost_value_scalar(spec_a));
  prost_value_struct_set(out, "device_rule_batches", prost_value_scalar(dev_batches));
  prost_value_struct_set(out, "arithmetic", prost_value_string(arith != 0 ? "fmad" : "exact"));   // the class the iteration kernels of this solve run in
  prost_value_struct_set(out, "iterations_per_launch_max", prost_value_scalar(gmax));             // tolerance class: largest group of iterations in one launch (0: pairs / singles)
  prost_value_struct_set(out, "operator_in_prox_kernels", prost_value_scalar(op_fused));      // generic PDHG: K x / K^T y formed inside the prox launches
  prost_value_struct_set(out, "residual_sums_in_prox_launches", prost_value_scalar(res_prox)); // generic PDHG, separate products: the prox launches add up the residual terms
  {   // sparse blocks applied from row patterns instead of their CSR arrays (forward + adjoint products counted separately)
    double pat = 0;
    for (const auto& blk : h.problem->linop()->blocks())
      if (auto* sb = dynamic_cast<BlockSparse<T>*>(blk.get())) pat += (sb->patterns_forward() ? 1 : 0) + (sb->patterns_adjoint() ? 1 : 0);
    prost_value_struct_set(out, "sparse_pattern_products", prost_value_scalar(pat));
  }
  if (nlhs >= 1) plhs[0] = out; else prost_value_free(out);
}
void cmd_solver_state(CMD_ARGS) {
  if (nrhs < 1) throw Exception("solver_state: handle [, with_vectors = 1] required.");
  AnyHandle& a = handle_of(prhs[0]);
  const bool vectors = nrhs < 2 || prhs[1]->data[0] > 0;      // 0: step sizes / residuals only (states too large to read back)
  if (a.single) solver_state_t(*std::static_pointer_cast<SolverHandle<float>>(a.h), vectors, nlhs, plhs);
  else solver_state_t(*std::static_pointer_cast<SolverHandle<double>>(a.h), vectors, nlhs, plhs);
}
// ---- column-sharded images: halo columns of the current iterate.  gradient2d with L channels keeps 3 L image planes of
// nl * ny entries: x (L planes) and y (2 L planes: d/dx of every channel, then d/dy); column c of a plane = ny contiguous
// entries at c * ny ----
template <typename T>
static BackendPDHG<T>& pdhg_of(AnyHandle& a) {
  auto h = std::static_pointer_cast<SolverHandle<T>>(a.h);
  auto* pd = dynamic_cast<BackendPDHG<T>*>(h->backend.get());
  if (!pd || !pd->sharded_path()) throw Exception("Halo exchange needs a pdhg solver on a one-kernel gradient2d path (L <= 4).");
  return *pd;
}
template <typename T>
static std::vector<T*> image_planes(BackendPDHG<T>& pd, size_t n) {
  const size_t L = pd.fused_channels(), P = n / L;
  std::vector<T*> planes;
  for (size_t l = 0; l < L; l++) planes.push_back(pd.x_data() + l * P);
  for (size_t k = 0; k < 2 * L; k++) planes.push_back(pd.y_data() + k * P);
  return planes;
}
/// solver_halo_exchange(handle, ny, halo, left_halo, right_halo, left_rank, right_rank): over the RCCL
/// communicator of comm_init.  left_halo / right_halo = number of halo columns this slab has on that side
/// (0 at the image border); a negative rank means no neighbour on that side.
template <typename T>
static void halo_exchange_t(AnyHandle& a, size_t ny, size_t H, size_t HL, size_t HR, int left, int right) {
  if (left < 0 && right < 0) return;
  if (!g_comm) throw Exception("solver_halo_exchange: comm_init first.");
  BackendPDHG<T>& pd = pdhg_of<T>(a);
  auto h = std::static_pointer_cast<SolverHandle<T>>(a.h);
  const size_t n = h->problem->ncols(), P = n / pd.fused_channels(), nl = P / ny;
  if (nl * ny != P || HL + HR + 2 * H > nl + (HL ? 0 : H) + (HR ? 0 : H)) throw Exception("solver_halo_exchange: slab too narrow for the halo width.");
  const size_t bytes = H * ny * sizeof(T);
  void* st = CurrentStream();
  CheckHip(prost_hip_comm_group_start(), "group_start");
  for (T* p : image_planes<T>(pd, n)) {
    if (left >= 0) {
      CheckHip(prost_hip_comm_send(g_comm, p + HL * ny, bytes, left, st), "send");              // my first owned columns
      CheckHip(prost_hip_comm_recv(g_comm, p, bytes, left, st), "recv");                        // -> my left halo
    }
    if (right >= 0) {
      CheckHip(prost_hip_comm_send(g_comm, p + (nl - HR - H) * ny, bytes, right, st), "send");  // my last owned columns
      CheckHip(prost_hip_comm_recv(g_comm, p + (nl - HR) * ny, bytes, right, st), "recv");      // -> my right halo
    }
  }
  CheckHip(prost_hip_comm_group_end(), "group_end");
}
/// solver_iterate_sharded(handle, iters, ny, halo, left_halo, right_halo, left_rank, right_rank, since_exchange) ->
/// since_exchange: `iters` iterations of a column slab with the halo refresh every halo - 2 iterations, all inside the
/// native solver (collective: every rank calls it with the same counts) -- no host language between the exchanges
template <typename T>
static size_t iterate_sharded_t(AnyHandle& a, int iters, size_t ny, size_t H, size_t HL, size_t HR, int left, int right, size_t since) {
  auto h = std::static_pointer_cast<SolverHandle<T>>(a.h);
  if (H < 3) throw Exception("solver_iterate_sharded: the halo must be at least 3 columns.");
  const size_t period = H - 2;
  // residual-driven rule on the device and a device-side transport (RCCL): the whole call is ONE sequence of enqueues -- the backend
  // calls the exchange from inside its batches (BackendPDHG::SetExchangeHook), the host waits once per batch of up to 240 iterations
  if (g_comm && !prost_hip_comm_is_host(g_comm) && (left >= 0 || right >= 0)) {
    if (auto* pd = dynamic_cast<BackendPDHG<T>*>(h->backend.get())) {
      if (pd->device_rules()) {
        pd->SetExchangeHook([&a, ny, H, HL, HR, left, right]() { halo_exchange_t<T>(a, ny, H, HL, HR, left, right); }, period, since);
        try { h->solver->Iterate(iters); } catch (...) { pd->ClearExchangeHook(); throw; }
        since = pd->since_exchange();
        pd->ClearExchangeHook();
        return since;
      }
    }
  }
  for (int done = 0; done < iters;) {
    if (since >= period) { halo_exchange_t<T>(a, ny, H, HL, HR, left, right); since = 0; }
    const int k = (int)std::min<size_t>((size_t)(iters - done), period - since);
    h->solver->Iterate(k);
    done += k; since += (size_t)k;
  }
  return since;
}
void cmd_solver_iterate_sharded(CMD_ARGS) {
  if (nrhs != 9) throw Exception("solver_iterate_sharded: (handle, iters, ny, halo, left_halo, right_halo, left_rank, right_rank, since_exchange) required.");
  AnyHandle& a = handle_of(prhs[0]);
  const int iters = (int)prhs[1]->data[0];
  const size_t ny = (size_t)prhs[2]->data[0], H = (size_t)prhs[3]->data[0], HL = (size_t)prhs[4]->data[0], HR = (size_t)prhs[5]->data[0];
  const int left = (int)prhs[6]->data[0], right = (int)prhs[7]->data[0];
  const size_t since = (size_t)prhs[8]->data[0];
  const size_t out = a.single ? iterate_sharded_t<float>(a, iters, ny, H, HL, HR, left, right, since) : iterate_sharded_t<double>(a, iters, ny, H, HL, HR, left, right, since);
  if (nlhs >= 1) plhs[0] = prost_value_scalar((double)out);
}
void cmd_solver_halo_exchange(CMD_ARGS) {
  (void)nlhs; (void)plhs;
  if (nrhs != 7) throw Exception("solver_halo_exchange: (handle, ny, halo, left_halo, right_halo, left_rank, right_rank) required.");
  AnyHandle& a = handle_of(prhs[0]);
  const size_t ny = (size_t)prhs[1]->data[0], H = (size_t)prhs[2]->data[0], HL = (size_t)prhs[3]->data[0], HR = (size_t)prhs[4]->data[0];
  const int left = (int)prhs[5]->data[0], right = (int)prhs[6]->data[0];
  if (a.single) halo_exchange_t<float>(a, ny, H, HL, HR, left, right); else halo_exchange_t<double>(a, ny, H, HL, HR, left, right);
}
/// solver_copy_columns(dst_handle, dst_col, src_handle, src_col, ncols, ny): the same transfer between two
/// solvers of ONE process (several slabs on one GPU: tests, or images larger than fit one solver's tiling)
template <typename T>
static void copy_columns_t(AnyHandle& dst, size_t dcol, AnyHandle& src, size_t scol, size_t ncols, size_t ny) {
  BackendPDHG<T>& pd = pdhg_of<T>(dst); BackendPDHG<T>& ps = pdhg_of<T>(src);
  const size_t nd = std::static_pointer_cast<SolverHandle<T>>(dst.h)->problem->ncols(), ns = std::static_pointer_cast<SolverHandle<T>>(src.h)->problem->ncols();
  if (pd.fused_channels() != ps.fused_channels()) throw Exception("solver_copy_columns: the slabs have different channel counts.");
  const size_t L = pd.fused_channels();
  if ((dcol + ncols) * ny > nd / L || (scol + ncols) * ny > ns / L) throw Exception("solver_copy_columns: column range outside the slab.");
  const std::vector<T*> d = image_planes<T>(pd, nd), s = image_planes<T>(ps, ns);
  for (size_t k = 0; k < d.size(); k++)
    CheckHip(prost_hip_memcpy_d2d(d[k] + dcol * ny, s[k] + scol * ny, ncols * ny * sizeof(T), CurrentStream()), "memcpy_d2d");
}
void cmd_solver_copy_columns(CMD_ARGS) {
  (void)nlhs; (void)plhs;
  if (nrhs != 6) throw Exception("solver_copy_columns: (dst_handle, dst_col, src_handle, src_col, ncols, ny) required.");
  AnyHandle& d = handle_of(prhs[0]); AnyHandle& s = handle_of(prhs[2]);
  if (d.single != s.single) throw Exception("solver_copy_columns: both solvers must have the same precision.");
  const size_t dcol = (size_t)prhs[1]->data[0], scol = (size_t)prhs[3]->data[0], nc = (size_t)prhs[4]->data[0], ny = (size_t)prhs[5]->data[0];
  if (d.single) copy_columns_t<float>(d, dcol, s, scol, nc, ny); else copy_columns_t<double>(d, dcol, s, scol, nc, ny);
}
// ---- verification without a full read-back (the 2048 x 2048 x 64 state is ~8 GB per solver) ----
/// solver_compare(handle_a, handle_b) -> 4 x 2 matrix, rows x, y, x_prev, y_prev: {elements that differ in value, sum |a - b|},
/// computed on the device (prost_hip_compare_*)
template <typename T>
static prost_value* solver_compare_t(AnyHandle& a, AnyHandle& b) {
  auto ha = std::static_pointer_cast<SolverHandle<T>>(a.h); auto hb = std::static_pointer_cast<SolverHandle<T>>(b.h);
  auto* pa = dynamic_cast<BackendPDHG<T>*>(ha->backend.get()); auto* pb = dynamic_cast<BackendPDHG<T>*>(hb->backend.get());
  if (!pa || !pb) throw Exception("solver_compare: both solvers must use the pdhg backend.");
  const size_t n = ha->problem->ncols(), m = ha->problem->nrows();
  if (n != hb->problem->ncols() || m != hb->problem->nrows()) throw Exception("solver_compare: problem sizes differ.");
  T* va[4]; T* vb[4];
  pa->device_iterates(va[0], va[1], va[2], va[3]);
  pb->device_iterates(vb[0], vb[1], vb[2], vb[3]);
  const size_t len[4] = {n, m, n, m};
  double* out = nullptr; void* ws = nullptr;
  CheckHip(prost_hip_malloc((void**)&out, 8 * sizeof(double)), "malloc");
  CheckHip(prost_hip_malloc(&ws, prost_hip_reduce_workspace_bytes()), "malloc");
  for (int k = 0; k < 4; k++) CheckHip(Api<T>::compare(out + 2 * k, va[k], vb[k], len[k], ws, CurrentStream()), "compare");
  double host[8];
  CheckHip(prost_hip_memcpy_d2h(host, out, sizeof(host), CurrentStream()), "d2h");
  CheckHip(prost_hip_stream_synchronize(CurrentStream()), "sync");
  prost_hip_free(out); prost_hip_free(ws);
  double t[8];
  for (int k = 0; k < 4; k++) { t[k] = host[2 * k]; t[4 + k] = host[2 * k + 1]; }      // column-major 4 x 2
  return prost_value_matrix(t, 4, 2);
}
void cmd_solver_compare(CMD_ARGS) {
  if (nrhs != 2) throw Exception("solver_compare: (handle_a, handle_b) required.");
  AnyHandle& a = handle_of(prhs[0]); AnyHandle& b = handle_of(prhs[1]);
  if (a.single != b.single) throw Exception("solver_compare: both solvers must have the same precision.");
  prost_value* out = a.single ? solver_compare_t<float>(a, b) : solver_compare_t<double>(a, b);
  if (nlhs >= 1) plhs[0] = out; else prost_value_free(out);
}
/// solver_read(handle, which, offsets, count) -> count x #offsets matrix: `count` consecutive entries of vector `which`
/// ("x", "y", "x_prev", "y_prev") starting at each offset
template <typename T>
static prost_value* solver_read_t(AnyHandle& a, const std::string& which, const prost_value* offs, size_t count) {
  auto h = std::static_pointer_cast<SolverHandle<T>>(a.h);
  auto* pd = dynamic_cast<BackendPDHG<T>*>(h->backend.get());
  if (!pd) throw Exception("solver_read: pdhg backend required.");
  T* v[4];
  pd->device_iterates(v[0], v[1], v[2], v[3]);
  const size_t n = h->problem->ncols(), m = h->problem->nrows();
  const int k = which == "x" ? 0 : which == "y" ? 1 : which == "x_prev" ? 2 : which == "y_prev" ? 3 : -1;
  if (k < 0) throw Exception("solver_read: vector must be one of x, y, x_prev, y_prev.");
  const size_t len = (k & 1) ? m : n, segs = offs->data.size();
  std::vector<T> buf(count * segs);
  for (size_t s = 0; s < segs; s++) {
    const size_t o = (size_t)offs->data[s];
    if (o + count > len) throw Exception("solver_read: segment outside the vector.");
    CheckHip(prost_hip_memcpy_d2h(buf.data() + s * count, v[k] + o, count * sizeof(T), CurrentStream()), "d2h");
  }
  CheckHip(prost_hip_stream_synchronize(CurrentStream()), "sync");
  prost_value* out = prost_value_matrix(nullptr, count, segs);
  for (size_t i = 0; i < buf.size(); i++) out->data[i] = (double)buf[i];
  return out;
}
void cmd_solver_read(CMD_ARGS) {
  if (nrhs != 4) throw Exception("solver_read: (handle, vector name, offsets, count) required.");
  AnyHandle& a = handle_of(prhs[0]);
  const std::string which = GetString(prhs[1]);
  if (prhs[2]->kind != PROST_VALUE_MATRIX) throw Exception("solver_read: offsets must be a numeric vector.");
  const size_t count = (size_t)prhs[3]->data[0];
  prost_value* out = a.single ? solver_read_t<float>(a, which, prhs[2], count) : solver_read_t<double>(a, which, prhs[2], count);
  if (nlhs >= 1) plhs[0] = out; else prost_value_free(out);
}
void cmd_solver_destroy(CMD_ARGS) {
  (void)nlhs; (void)plhs;
  if (nrhs < 1) throw Exception("solver_destroy: handle required.");
  AnyHandle& a = handle_of(prhs[0]);
  if (a.single) std::static_pointer_cast<SolverHandle<float>>(a.h)->solver->Release();
  else std::static_pointer_cast<SolverHandle<double>>(a.h)->solver->Release();
  g_handles.erase((int)prhs[0]->data[0]);
}

void cmd_comm_unique_id(CMD_ARGS) {
  (void)nrhs; (void)prhs;
  unsigned char id[128];
  CheckHip(prost_hip_comm_unique_id(id), "comm_unique_id");
  std::vector<double> d(id, id + 128);
  if (nlhs >= 1) plhs[0] = prost_value_matrix(d.data(), 1, 128);
}
void cmd_comm_init(CMD_ARGS) {
  (void)nlhs; (void)plhs;
  if (nrhs != 3 || prhs[0]->data.size() != 128) throw Exception("comm_init: (id[128], rank, world) required.");
  select_device();
  unsigned char id[128];
  for (int i = 0; i < 128; i++) id[i] = (unsigned char)prhs[0]->data[i];
  if (g_comm) { prost_hip_comm_destroy(g_comm); g_comm = nullptr; }
  CheckHip(prost_hip_comm_create(&g_comm, id, (int)prhs[1]->data[0], (int)prhs[2]->data[0]), "comm_create");
  g_comm_world = (int)prhs[2]->data[0];
}
/// comm_info() -> struct {nranks, transport}: nranks as the communicator itself counts them (ncclCommCount), 0 / "none"
/// without a communicator
void cmd_comm_info(CMD_ARGS) {
  (void)nrhs; (void)prhs;
  prost_value* out = prost_value_struct();
  int n = 0;
  if (g_comm) CheckHip(prost_hip_comm_count(g_comm, &n), "comm_count");
  prost_value_struct_set(out, "nranks", prost_value_scalar((double)n));
  prost_value_struct_set(out, "transport", prost_value_string(!g_comm ? "none" : prost_hip_comm_is_host(g_comm) ? "host" : "rccl"));
  if (nlhs >= 1) plhs[0] = out; else prost_value_free(out);
}
void cmd_comm_destroy(CMD_ARGS) {
  (void)nlhs; (void)plhs; (void)nrhs; (void)prhs;
  if (g_comm) { prost_hip_comm_destroy(g_comm); g_comm = nullptr; g_comm_world = 1; }
}
/// load_plugin(path): dlopen()s a shared library of user-defined prost::Block / prost::Prox / prost::Backend subclasses whose
/// static initialisers insert `name -> factory` into Factory<T>::block_reg() / prox_reg() / backend_reg() -- what compiling
/// user sources into the MEX file does for the reference (custom.cpp:11-28, cmake/CustomSources.cmake.example:1-26).  The
/// library stays loaded for the life of the process.
void cmd_load_plugin(CMD_ARGS) {
  (void)nlhs; (void)plhs;
  if (nrhs != 1) throw Exception("load_plugin: (path) required.");
  const std::string path = GetString(prhs[0]);
  if (!dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL)) {
    const char* why = dlerror();
    throw Exception("load_plugin: cannot load '" + path + "': " + (why ? why : "unknown error"));
  }
}
/// registered() -> struct {prox, block, backend}: cells of the names in the registries of the current precision
void cmd_registered(CMD_ARGS) {
  (void)nrhs; (void)prhs;
  auto names = [](auto& reg) {
    prost_value* c = prost_value_cell(reg.size());
    size_t i = 0;
    for (auto& e : reg) prost_value_cell_set(c, i++, prost_value_string(e.first.c_str()));
    return c;
  };
  prost_value* out = prost_value_struct();
  if (g_single) {
    prost_value_struct_set(out, "prox", names(Factory<float>::prox_reg())); prost_value_struct_set(out, "block", names(Factory<float>::block_reg()));
    prost_value_struct_set(out, "backend", names(Factory<float>::backend_reg()));
  } else {
    prost_value_struct_set(out, "prox", names(Factory<double>::prox_reg())); prost_value_struct_set(out, "block", names(Factory<double>::block_reg()));
    prost_value_struct_set(out, "backend", names(Factory<double>::backend_reg()));
  }
  if (nlhs >= 1) plhs[0] = out; else prost_value_free(out);
}
void cmd_set_quirks(CMD_ARGS) {
  (void)nlhs; (void)plhs;
  if (nrhs < 1) throw Exception("set_quirks: struct required.");
  if (prost_value_field(prhs[0], "diags_adjoint_grid")) BlockDiags<double>::SetReferenceGridQuirk(GetScalarFromField(prhs[0], "diags_adjoint_grid") > 0);
  if (prost_value_field(prhs[0], "fuse_moreau")) { const bool on = GetScalarFromField(prhs[0],
      "fuse_moreau") > 0; ProxMoreau<float>::SetFuseElemOperations(on); ProxMoreau<double>::SetFuseElemOperations(on); }
  if (prost_value_field(prhs[0], "sparse_patterns")) BlockSparse<double>::SetPatternCompression(GetScalarFromField(prhs[0], "sparse_patterns") > 0);
  if (prost_value_field(prhs[0], "sparse_stencils")) BlockSparse<double>::SetStencilRecognition(GetScalarFromField(prhs[0], "sparse_stencils") > 0);
  if (prost_value_field(prhs[0], "dual_negate_float")) DualLinearOperator<double>::SetReferenceNegateQuirk(GetScalarFromField(prhs[0], "dual_negate_float") > 0);
}

}  // namespace
extern "C" int prost_comm_init_host(prost_allreduce_cb fn, void* user, int world_size) {
  try {
    select_device();
    if (g_comm) { prost_hip_comm_destroy(g_comm); g_comm = nullptr; }
    CheckHip(prost_hip_comm_create_host(&g_comm, fn, user), "comm_create_host");
    CheckHip(prost_hip_comm_host_configure(g_comm, world_size, nullptr, nullptr), "comm_host_configure");
    g_comm_world = world_size;
    return 0;
  } catch (const std::exception& e) { g_error = e.what(); return 1; }
}
extern "C" int prost_comm_set_host_p2p(prost_p2p_cb fn, void* user) {
  try {
    if (!g_comm || !prost_hip_comm_is_host(g_comm)) throw Exception("prost_comm_set_host_p2p: prost_comm_init_host first.");
    CheckHip(prost_hip_comm_host_configure(g_comm, g_comm_world, fn, user), "comm_host_configure");
    return 0;
  } catch (const std::exception& e) { g_error = e.what(); return 1; }
}
namespace {
typedef void (*cmd_fn)(CMD_ARGS);
const std::map<std::string, cmd_fn>& cmd_reg() {
  static const std::map<std::string, cmd_fn> reg = {
      {"init", cmd_init}, {"release", cmd_release}, {"solve_problem", cmd_solve_problem}, {"eval_linop", cmd_eval_linop},
      {"eval_prox", cmd_eval_prox}, {"list_gpus", cmd_list_gpus}, {"set_gpu", cmd_set_gpu},
      {"set_precision", cmd_set_precision}, {"get_precision", cmd_get_precision}, {"problem_info", cmd_problem_info}, {"glibc_rand_unit", cmd_glibc_rand_unit},
      {"solver_create", cmd_solver_create}, {"solver_iterate", cmd_solver_iterate}, {"solver_kernel_times", cmd_solver_kernel_times}, {"solver_state", cmd_solver_state},
      {"solver_destroy", cmd_solver_destroy}, {"solver_halo_exchange", cmd_solver_halo_exchange}, {"solver_iterate_sharded",
          cmd_solver_iterate_sharded}, {"solver_copy_columns", cmd_solver_copy_columns},
      {"solver_compare", cmd_solver_compare}, {"solver_read", cmd_solver_read}, {"comm_unique_id", cmd_comm_unique_id}, {"comm_init", cmd_comm_init},
      {"comm_destroy", cmd_comm_destroy}, {"comm_info", cmd_comm_info}, {"set_quirks", cmd_set_quirks}, {"load_plugin", cmd_load_plugin}, {"registered", cmd_registered}};
  return reg;
}

}  // namespace

extern "C" int prost_command(const char* cmd, int nlhs, prost_value** plhs, int nrhs, const prost_value* const* prhs) {
  for (int i = 0; i < nlhs; i++) plhs[i] = nullptr;
  try {
    if (!cmd) throw Exception("Usage: prost_(command, arg1, arg2, ...);");
    auto it = cmd_reg().find(cmd);
    if (it == cmd_reg().end()) { std::stringstream msg; msg << "Unknown command '" << cmd << "'."; throw Exception(msg.str()); }
    for (int i = 0; i < nrhs; i++) if (!prhs[i]) throw Exception("Null argument passed to prost_command.");
    it->second(nlhs, plhs, nrhs, prhs);
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    for (int i = 0; i < nlhs; i++) { if (plhs[i]) { prost_value_free(plhs[i]); plhs[i] = nullptr; } }
    return 1;
  }
}
