// CPU harness for the ENQUEUE ORDER of a column-sharded slab whose residual-driven step-size rule runs on the device (round 6;
// tests/test_host_schedule.py).  The host sources of the solver are compiled into this translation unit as they are; the kernel C ABI
// (include/prost_hip.h) is replaced by a recording mock: every entry point the PDHG backend reaches on this path allocates host
// memory or appends a line to a log -- nothing is computed.  What is checked is the schedule a real 8-GPU run will enqueue on each
// rank's stream: iteration kernel (+ partial sums) -> all-reduce of the four sums -> rule kernel -> [halo exchange when due] -> next
// iteration kernel, no launch across an exchange, ONE host wait per batch.  Entry points that are not on this path stay unresolved
// (-Wl,--unresolved-symbols=ignore-all).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include "prost_hip.h"

static std::vector<std::string> g_log;
static void logf(const char* fmt, ...) {
  char buf[256];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  g_log.push_back(buf);
}
static prost_hip_pdhg_rule_state g_rule;      // what the rule kernels would keep on the device

extern "C" {
const char* prost_hip_last_error(void) { return ""; }
int prost_hip_check_last_error(void) { return 0; }
int prost_hip_malloc(void** p, size_t bytes) { *p = calloc(bytes ? bytes : 1, 1); return 0; }
int prost_hip_free(void* p) { free(p); return 0; }
int prost_hip_host_alloc(void** p, size_t bytes) { *p = calloc(bytes ? bytes : 1, 1); return 0; }
int prost_hip_host_free(void* p) { free(p); return 0; }
int prost_hip_memcpy_h2d(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
int prost_hip_memcpy_d2h(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
int prost_hip_memcpy_d2d(void* d, const void* s, size_t n, void*) { memmove(d, s, n); return 0; }
int prost_hip_memset(void* d, int v, size_t n, void*) { memset(d, v, n); return 0; }
int prost_hip_stream_create(void** s) { *s = (void*)0x10; return 0; }
int prost_hip_stream_destroy(void*) { return 0; }
int prost_hip_stream_synchronize(void*) { logf("HOST WAIT (stream)"); return 0; }
int prost_hip_device_synchronize(void) { logf("HOST WAIT (device)"); return 0; }
int prost_hip_event_create(void** e) { *e = malloc(1); return 0; }
int prost_hip_event_create_timing(void** e) { *e = malloc(1); return 0; }
int prost_hip_event_destroy(void* e) { free(e); return 0; }
int prost_hip_event_record(void*, void*) { return 0; }
int prost_hip_event_synchronize(void*) { logf("HOST WAIT (event)"); return 0; }
int prost_hip_stream_wait_event(void*, void*) { return 0; }
int prost_hip_next_launch_events(void*, void*) { return 0; }
size_t prost_hip_reduce_workspace_bytes(void) { return 1 << 16; }
size_t prost_hip_pdhg_rule_record_bytes(void) { return 4096; }
int prost_hip_pdhg_record_view(const void* rec, int, const void** tau, const void** sigma, const void** theta, const int** stop) {
  if (tau) *tau = rec; if (sigma) *sigma = (char*)rec + 8; if (theta) *theta = (char*)rec + 16; if (stop) *stop = (const int*)((char*)rec + 24);
  return 0;
}
// ---- what the fused gray-value path asks about a description
int prost_hip_fused_supported(const prost_hip_fused_desc*, int) { return 1; }
int prost_hip_fused_iteration_supported(const prost_hip_fused_desc*, int) { return 1; }
int prost_hip_fused_iteration2_supported(const prost_hip_fused_desc*, int) { return 1; }
int prost_hip_fused_iteration2_profitable(const prost_hip_fused_desc*, int) { return 1; }
int prost_hip_fused_iteration2_chunk_cols(const prost_hip_fused_desc*, int, int) { return 18; }
int prost_hip_fused_iterationk_max(const prost_hip_fused_desc*, int) { return 0; }
int prost_hip_fused_iteration3d_supported(const prost_hip_fused_desc*, int) { return 0; }
int prost_hip_fused_iteration3d_pw_supported(const prost_hip_fused_desc*, int) { return 0; }
int prost_hip_fused_iteration3d_x2_supported(const prost_hip_fused_desc*, int) { return 0; }
int prost_hip_fused_iteration_mc_supported(const prost_hip_fused_desc*, int) { return 0; }
int prost_hip_fused_iteration_mc_x2_profitable(const prost_hip_fused_desc*, int) { return 0; }
int prost_hip_mem_info(size_t* f, size_t* t) { *f = *t = (size_t)1 << 34; return 0; }
int prost_hip_get_device(int* d) { *d = 0; return 0; }
int prost_hip_fill_f32(float* p, double v, size_t n, void*) { for (size_t i = 0; i < n; i++) p[i] = (float)v; return 0; }
int prost_hip_event_elapsed_ms(void*, void*, float* ms) { *ms = 0; return 0; }
int prost_hip_comm_is_host(void*) { return getenv("MOCK_HOST_TRANSPORT") ? 1 : 0; }
// ---- the launches of the path
int prost_hip_fused_iteration_f32(const prost_hip_fused_desc*, float*, float*, const float*, const float*, const float*, double, double, double, int, int, int, int,
                                  double* res, void*, void*) {
  logf("iteration x1 %s(host-side step sizes)", res ? "+sums " : ""); return 0;
}
int prost_hip_fused_iteration2_f32(const prost_hip_fused_desc*, float*, float*, const float*, const float*, float* xm, float*, const double*, const double*, const double*, int,
                                   double* res, void*, void*) {
  logf("iteration x2 %s%s(host-side step sizes)", res ? "+sums " : "", xm ? "+mid " : ""); return 0;
}
int prost_hip_pdhg_rule_begin_f32(void*, const prost_hip_pdhg_rule_opts*, const prost_hip_fused_desc*, double tau, double sigma, double theta, double alpha, int l, int u,
                                  int stop_on_convergence, prost_hip_pdhg_rule_state* mirror, void*) {
  memset(&g_rule, 0, sizeof(g_rule));
  g_rule.tau = g_rule.prev_tau = tau; g_rule.sigma = g_rule.prev_sigma = sigma; g_rule.theta = g_rule.prev_theta = theta; g_rule.arg_alpha = alpha; g_rule.arb_l = l; g_rule.arb_u = u;
  if (mirror) *mirror = g_rule;
  logf("rule_begin (stop_on_convergence %d)", stop_on_convergence); return 0;
}
static void evaluate(unsigned long long iteration, prost_hip_pdhg_rule_state* mirror) {
  g_rule.evaluations++; g_rule.last_iteration = iteration; g_rule.primal_res = g_rule.dual_res = 1.0;
  if (mirror) *mirror = g_rule;
}
int prost_hip_fused_iteration_rec_f32(const prost_hip_fused_desc*, float*, float*, const float*, const float*, const float*, void*, int, int, int, int, double* res, void*,
                                      int apply_rule, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void*) {
  logf("iteration x1 k=%llu%s%s", iteration, res ? " +sums" : "", res && apply_rule ? " +rule" : "");
  if (res && apply_rule) evaluate(iteration, mirror);
  return 0;
}
int prost_hip_fused_iteration2_rec_f32(const prost_hip_fused_desc*, float*, float*, const float*, const float*, float* xm, float*, void*, int, double* res, void*, int apply_rule,
                                       unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void*) {
  logf("iteration x2 k=%llu,%llu%s%s%s", iteration - 1, iteration, xm ? " +mid" : "", res ? " +sums" : "", res && apply_rule ? " +rule" : "");
  if (res && apply_rule) evaluate(iteration, mirror);
  return 0;
}
int prost_hip_allreduce_sum_f64(void*, double*, size_t n, void*) { logf("all-reduce of %zu sums", n); return 0; }
int prost_hip_pdhg_rule_apply_f32(void*, const double*, unsigned long long iteration, prost_hip_pdhg_rule_state* mirror, void*) {
  logf("rule kernel k=%llu", iteration); evaluate(iteration, mirror); return 0;
}
}  // extern "C"

#include "../../prost_amd/csrc/host/common.cpp"
#include "../../prost_amd/csrc/host/linop.cpp"
#include "../../prost_amd/csrc/host/prox.cpp"
#include "../../prost_amd/csrc/host/problem.cpp"
#include "../../prost_amd/csrc/host/backend_pdhg.cpp"
#include "../../prost_amd/csrc/host/solver.cpp"

using namespace prost;

// a call through an entry point this mock does not define jumps to address 0: say where from (build with -rdynamic)
static void on_segv(int) {
  void* frames[32];
  const int n = backtrace(frames, 32);
  const char msg[] = "slab_schedule_harness: call of an entry point the mock does not define (or a crash); backtrace:\n";
  if (write(2, msg, sizeof(msg) - 1) < 0) _exit(3);
  backtrace_symbols_fd(frames, n, 2);
  _exit(3);
}

int main(int argc, char** argv) {
  signal(SIGSEGV, on_segv);
  const int iters = argc > 1 ? atoi(argv[1]) : 40, residual_iter = argc > 2 ? atoi(argv[2]) : 3, halo = argc > 3 ? atoi(argv[3]) : 8;
  const bool pairs = !(argc > 4 && atoi(argv[4]) == 0);
  const size_t nx = 64, ny = 32, n = nx * ny;
  auto problem = std::make_shared<Problem<float>>();
  problem->SetDimensions(2 * n, n);
  problem->SetScalingAlpha(1);
  problem->AddBlock(std::make_shared<BlockGradient2D<float>>(0, 0, nx, ny, 1, false));
  std::array<std::vector<float>, 7> cg = {{{1.f}, std::vector<float>(n, 0.5f), {10.f}, {0.f}, {0.f}, {0.f}, {0.f}}};
  std::array<std::vector<float>, 7> cf = {{{1.f}, {1.f}, {1.f}, {0.f}, {0.f}, {0.f}, {0.f}}};
  problem->AddProx_g(std::make_shared<ProxElemDispatch<float>>(PROST_OP_1D, PROST_FN_SQUARE, 0, n, 1, false, true, cg));
  problem->AddProx_fstar(std::make_shared<ProxElemDispatch<float>>(PROST_OP_NORM2, PROST_FN_IND_LEQ0, 0, n, 2, false, false, cf));
  BackendPDHG<float>::Options bo;
  bo.stepsize_variant = BackendPDHG<float>::kPDHGStepsResidualBoyd; bo.residual_iter = residual_iter; bo.scale_steps_operator = false;
  bo.allow_pair_kernel = pairs;
  auto backend = std::make_shared<BackendPDHG<float>>(bo);
  backend->SetCommunicator((void*)0x1, 2 * 2 * n, 2 * n);           // two slabs
  backend->SetOwnedColumns((size_t)halo, nx - (size_t)halo);
  Solver<float> solver(problem, backend);
  Solver<float>::Options so; so.tol_rel_primal = so.tol_rel_dual = so.tol_abs_primal = so.tol_abs_dual = 0; so.max_iters = 1 << 20; so.num_cback_calls = 0;
  solver.SetOptions(so);
  solver.Initialize();
  std::printf("path %s device_rules %d\n", backend->path().c_str(), (int)backend->device_rules());
  g_log.clear();
  int exchanges = 0;
  backend->SetExchangeHook([&]() { exchanges++; logf("HALO EXCHANGE"); }, (size_t)(halo - 2), 0);
  solver.Iterate(iters);
  std::printf("iterations %zu since_exchange %zu exchanges %d\n", backend->iteration(), backend->since_exchange(), exchanges);
  for (const std::string& l : g_log) std::printf("%s\n", l.c_str());
  return 0;
}
