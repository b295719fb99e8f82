// runtime.hip -- device / memory / stream / event plumbing of the C ABI, plus the RCCL
// communicator used for the global stopping criterion of multi-GPU batches.
#include <cstdlib>
#include "common.hpp"

#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

namespace prost_hip {

static thread_local std::string g_last_error;

thread_local hipEvent_t g_launch_ev_start = nullptr, g_launch_ev_stop = nullptr;
thread_local void* g_step_record = nullptr;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(hipError_t e, const char* what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return (int)e == 0 ? 1 : (int)e;
}

}  // namespace prost_hip

using namespace prost_hip;

extern "C" {

const char* prost_hip_last_error(void) { return g_last_error.c_str(); }
int prost_hip_abi_version(void) { return PROST_HIP_ABI_VERSION; }

int prost_hip_device_count(int* count) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { *count = 0; return fail(e, "hipGetDeviceCount"); }
  *count = c;
  return 0;
}
int prost_hip_set_device(int device) { PH_CHECK(hipSetDevice(device)); return 0; }
int prost_hip_get_device(int* device) { PH_CHECK(hipGetDevice(device)); return 0; }
int prost_hip_device_info(int device, char* name, size_t len, int* cu_count, size_t* total_mem) {
  hipDeviceProp_t prop;
  PH_CHECK(hipGetDeviceProperties(&prop, device));
  if (name && len) { std::strncpy(name, prop.name, len - 1); name[len - 1] = 0; }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (total_mem) *total_mem = prop.totalGlobalMem;
  return 0;
}
int prost_hip_mem_info(size_t* f, size_t* t) { PH_CHECK(hipMemGetInfo(f, t)); return 0; }
int prost_hip_malloc(void** p, size_t bytes) { *p = nullptr; if (bytes == 0) return 0; PH_CHECK(hipMalloc(p, bytes)); return 0; }
int prost_hip_free(void* p) { if (p) PH_CHECK(hipFree(p)); return 0; }
int prost_hip_host_alloc(void** p, size_t bytes) { PH_CHECK(hipHostMalloc(p, bytes, hipHostMallocDefault)); return 0; }
int prost_hip_host_free(void* p) { if (p) PH_CHECK(hipHostFree(p)); return 0; }
int prost_hip_memcpy_h2d(void* d, const void* s, size_t n, void* st) { if (n) PH_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, as_stream(st))); return 0; }
int prost_hip_memcpy_d2h(void* d, const void* s, size_t n, void* st) { if (n) PH_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, as_stream(st))); return 0; }
int prost_hip_memcpy_d2d(void* d, const void* s, size_t n, void* st) { if (n) PH_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, as_stream(st))); return 0; }
int prost_hip_memset(void* d, int v, size_t n, void* st) { if (n) PH_CHECK(hipMemsetAsync(d, v, n, as_stream(st))); return 0; }
int prost_hip_stream_create(void** s) { hipStream_t st; PH_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); *s = st; return 0; }
int prost_hip_stream_destroy(void* s) { if (s) PH_CHECK(hipStreamDestroy(as_stream(s))); return 0; }
int prost_hip_stream_synchronize(void* s) { PH_CHECK(hipStreamSynchronize(as_stream(s))); return 0; }
int prost_hip_device_synchronize(void) { PH_CHECK(hipDeviceSynchronize()); return 0; }
int prost_hip_event_create(void** e) { hipEvent_t ev; PH_CHECK(hipEventCreate(&ev)); *e = ev; return 0; }
int prost_hip_event_create_timing(void** e) {
  static const bool fence = []() { const char* v = getenv("PROST_TIMING_EVENTS_FENCE"); return v && atoi(v) != 0; }();      // A/B
  hipEvent_t ev; PH_CHECK(hipEventCreateWithFlags(&ev, fence ? hipEventDefault : hipEventDisableSystemFence)); *e = ev; return 0;
}
int prost_hip_event_destroy(void* e) { if (e) PH_CHECK(hipEventDestroy((hipEvent_t)e)); return 0; }
int prost_hip_event_record(void* e, void* s) { PH_CHECK(hipEventRecord((hipEvent_t)e, as_stream(s))); return 0; }
int prost_hip_stream_wait_event(void* s, void* e) { PH_CHECK(hipStreamWaitEvent(as_stream(s), (hipEvent_t)e, 0)); return 0; }
int prost_hip_event_synchronize(void* e) { PH_CHECK(hipEventSynchronize((hipEvent_t)e)); return 0; }
int prost_hip_use_step_record(void* record) { g_step_record = record; return 0; }
int prost_hip_next_launch_events(void* start, void* stop) {
  if (start != nullptr && stop == nullptr) { set_error("prost_hip_next_launch_events: a start event needs a stop event"); return 1; }
  g_launch_ev_start = (hipEvent_t)start; g_launch_ev_stop = (hipEvent_t)stop;
  return 0;
}
int prost_hip_event_elapsed_ms(void* a, void* b, float* ms) { PH_CHECK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b)); return 0; }
// ---- graphs: a captured launch sequence replayed with one host call ----
int prost_hip_stream_begin_capture(void* s) { PH_CHECK(hipStreamBeginCapture(as_stream(s), hipStreamCaptureModeThreadLocal)); return 0; }
int prost_hip_stream_end_capture(void* s, void** exec) {
  hipGraph_t g = nullptr;
  PH_CHECK(hipStreamEndCapture(as_stream(s), &g));
  hipGraphExec_t e = nullptr;
  const hipError_t r = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (r != hipSuccess) return fail(r, "hipGraphInstantiate");
  *exec = e;
  return 0;
}
int prost_hip_graph_launch(void* exec, void* s) { PH_CHECK(hipGraphLaunch((hipGraphExec_t)exec, as_stream(s))); return 0; }
int prost_hip_graph_destroy(void* exec) { if (exec) PH_CHECK(hipGraphExecDestroy((hipGraphExec_t)exec)); return 0; }
int prost_hip_check_last_error(void) { PH_CHECK(hipGetLastError()); return 0; }

// ---- communicators: RCCL over xGMI, or a host-callback transport ----
// A communicator handle is a Comm*: either an RCCL communicator (one rank per GPU, the production path) or a HOST
// transport whose all-reduce is a caller-supplied function working on pinned host memory (gloo / MPI on the host).  The
// host transport keeps the stream semantics of the RCCL call -- it is ENQUEUED (D2H copy, host function, H2D copy) and
// ordered by the stream like any other work -- so the callers' stream / event choreography runs unchanged; it exists to
// run the multi-rank logic where RCCL cannot (several ranks on ONE GPU: tests on a single-GPU box).
struct Comm {
  ncclComm_t nccl = nullptr;
  prost_hip_host_allreduce_fn host_fn = nullptr;
  void* host_user = nullptr;
  double* staging = nullptr;      // pinned, kHostStaging doubles
  size_t count = 0;               // doubles of the call in flight (read by the host function)
  // host transport, point-to-point: `p2p_fn` performs ALL transfers of one group (pinned host buffers) and returns when
  // they are complete; the staging area is reused by consecutive groups, which stream order keeps apart
  int host_world = 1;
  prost_hip_host_p2p_fn p2p_fn = nullptr;
  void* p2p_user = nullptr;
  char* p2p_staging = nullptr;    // pinned
  size_t p2p_capacity = 0;
};
// one group of sends / receives on a host communicator, handed to the host function through the stream
struct HostGroupJob {
  Comm* comm;
  std::vector<int> is_send, peers;
  std::vector<void*> bufs;        // pinned staging slices
  std::vector<size_t> bytes;
};
struct HostPendingOp { Comm* comm; bool send; void* dev; size_t bytes; int peer; hipStream_t stream; };
static thread_local int g_group_depth = 0;
static thread_local std::vector<HostPendingOp> g_pending;
static void host_p2p_trampoline(void* p) {
  HostGroupJob* j = static_cast<HostGroupJob*>(p);
  j->comm->p2p_fn(j->comm->p2p_user, (int)j->is_send.size(), j->is_send.data(), j->peers.data(), j->bufs.data(), j->bytes.data());
  delete j;
}
// Enqueues the pending operations of one host communicator as ONE exchange: D2H copies of everything that is sent, the host
// function (all transfers of the group, so both neighbours are served whatever order they post in), H2D copies of
// everything received.  Same enqueue-and-return contract as an RCCL group.
static int flush_host_group() {
  std::vector<HostPendingOp> ops;
  ops.swap(g_pending);
  if (ops.empty()) return 0;
  Comm* c = ops[0].comm;
  hipStream_t s = ops[0].stream;
  size_t total = 0;
  for (const HostPendingOp& o : ops) {
    if (o.comm != c || o.stream != s) { set_error("prost_hip_comm_group_end: one host communicator and one stream per group"); return 1; }
    total += (o.bytes + 63) & ~(size_t)63;
  }
  if (total > c->p2p_capacity) {
    // a previous group may still be using the old area: drain the stream before it is replaced
    PH_CHECK(hipStreamSynchronize(s));
    if (c->p2p_staging) (void)hipHostFree(c->p2p_staging);
    c->p2p_staging = nullptr; c->p2p_capacity = 0;
    PH_CHECK(hipHostMalloc((void**)&c->p2p_staging, total, hipHostMallocDefault));
    c->p2p_capacity = total;
  }
  HostGroupJob* job = new HostGroupJob;
  job->comm = c;
  size_t off = 0;
  for (const HostPendingOp& o : ops) {
    char* h = c->p2p_staging + off;
    off += (o.bytes + 63) & ~(size_t)63;
    job->is_send.push_back(o.send ? 1 : 0); job->peers.push_back(o.peer); job->bufs.push_back(h); job->bytes.push_back(o.bytes);
    if (o.send) { const hipError_t e = hipMemcpyAsync(h, o.dev, o.bytes, hipMemcpyDeviceToHost, s); if (e != hipSuccess) { delete job; return fail(e, "hipMemcpyAsync"); } }
  }
  const std::vector<void*> bufs = job->bufs;            // the job is deleted by the host function
  { const hipError_t e = hipLaunchHostFunc(s, host_p2p_trampoline, job); if (e != hipSuccess) { delete job; return fail(e, "hipLaunchHostFunc"); } }
  for (size_t i = 0; i < ops.size(); i++)
    if (!ops[i].send) PH_CHECK(hipMemcpyAsync(ops[i].dev, bufs[i], ops[i].bytes, hipMemcpyHostToDevice, s));
  return 0;
}
static int host_p2p(Comm* c, bool send, void* dev, size_t bytes, int peer, void* stream, const char* what) {
  if (!c->p2p_fn) { set_error(std::string(what) + ": this host-callback communicator has no point-to-point function (prost_hip_comm_host_configure)"); return 1; }
  g_pending.push_back({c, send, dev, bytes, peer, as_stream(stream)});
  return g_group_depth > 0 ? 0 : flush_host_group();
}
constexpr size_t kHostStaging = 64;
static void host_allreduce_trampoline(void* p) {
  Comm* c = static_cast<Comm*>(p);
  c->host_fn(c->host_user, c->staging, c->count);
}
static int nccl_fail(ncclResult_t r, const char* what) {
  set_error(std::string(what) + ": " + ncclGetErrorString(r));
  return 1000 + (int)r;
}
int prost_hip_comm_unique_id(void* id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId");
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}
int prost_hip_comm_create(void** comm, const void* id128, int rank, int world) {
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t c;
  ncclResult_t r = ncclCommInitRank(&c, world, id, rank);
  if (r != ncclSuccess) return nccl_fail(r, "ncclCommInitRank");
  Comm* h = new Comm;
  h->nccl = c;
  *comm = h;
  return 0;
}
int prost_hip_comm_create_host(void** comm, prost_hip_host_allreduce_fn fn, void* user) {
  if (!fn) { set_error("prost_hip_comm_create_host: all-reduce function required"); return 1; }
  Comm* h = new Comm;
  h->host_fn = fn; h->host_user = user;
  const hipError_t e = hipHostMalloc((void**)&h->staging, kHostStaging * sizeof(double), hipHostMallocDefault);
  if (e != hipSuccess) { delete h; return fail(e, "hipHostMalloc"); }
  *comm = h;
  return 0;
}
int prost_hip_comm_host_configure(void* comm, int world_size, prost_hip_host_p2p_fn p2p, void* p2p_user) {
  Comm* h = static_cast<Comm*>(comm);
  if (!h || h->nccl) { set_error("prost_hip_comm_host_configure: host-callback communicator required"); return 1; }
  if (world_size < 1) { set_error("prost_hip_comm_host_configure: world_size must be positive"); return 1; }
  h->host_world = world_size; h->p2p_fn = p2p; h->p2p_user = p2p_user;
  return 0;
}
int prost_hip_comm_count(void* comm, int* nranks) {
  Comm* h = static_cast<Comm*>(comm);
  if (!h) { set_error("prost_hip_comm_count: no communicator"); return 1; }
  if (!h->nccl) { *nranks = h->host_world; return 0; }
  ncclResult_t r = ncclCommCount(h->nccl, nranks);
  if (r != ncclSuccess) return nccl_fail(r, "ncclCommCount");
  return 0;
}
int prost_hip_comm_is_host(void* comm) { return comm && !static_cast<Comm*>(comm)->nccl ? 1 : 0; }
int prost_hip_comm_destroy(void* comm) {
  if (!comm) return 0;
  Comm* h = static_cast<Comm*>(comm);
  int rc = 0;
  if (h->nccl) { ncclResult_t r = ncclCommDestroy(h->nccl); if (r != ncclSuccess) rc = nccl_fail(r, "ncclCommDestroy"); }
  if (h->staging) (void)hipHostFree(h->staging);
  if (h->p2p_staging) (void)hipHostFree(h->p2p_staging);
  delete h;
  return rc;
}
// point-to-point over xGMI (halo columns of column-sharded images); calls between group_start / group_end
// are issued as one RCCL group, so a rank can send to and receive from both neighbours without deadlock
int prost_hip_comm_group_start(void) {
  ncclResult_t r = ncclGroupStart();
  if (r != ncclSuccess) return nccl_fail(r, "ncclGroupStart");
  g_group_depth++;
  return 0;
}
int prost_hip_comm_group_end(void) {
  if (g_group_depth > 0) g_group_depth--;
  ncclResult_t r = ncclGroupEnd();
  if (r != ncclSuccess) { g_pending.clear(); return nccl_fail(r, "ncclGroupEnd"); }
  return g_group_depth == 0 ? flush_host_group() : 0;
}
int prost_hip_comm_send(void* comm, const void* buf, size_t bytes, int peer, void* stream) {
  if (!static_cast<Comm*>(comm)->nccl) return host_p2p(static_cast<Comm*>(comm), true, const_cast<void*>(buf), bytes, peer, stream, "prost_hip_comm_send");
  ncclResult_t r = ncclSend(buf, bytes, ncclChar, peer, static_cast<Comm*>(comm)->nccl, as_stream(stream));
  if (r != ncclSuccess) return nccl_fail(r, "ncclSend");
  return 0;
}
int prost_hip_comm_recv(void* comm, void* buf, size_t bytes, int peer, void* stream) {
  if (!static_cast<Comm*>(comm)->nccl) return host_p2p(static_cast<Comm*>(comm), false, buf, bytes, peer, stream, "prost_hip_comm_recv");
  ncclResult_t r = ncclRecv(buf, bytes, ncclChar, peer, static_cast<Comm*>(comm)->nccl, as_stream(stream));
  if (r != ncclSuccess) return nccl_fail(r, "ncclRecv");
  return 0;
}
int prost_hip_allreduce_sum_f64(void* comm, double* buf, size_t count, void* stream) {
  Comm* h = static_cast<Comm*>(comm);
  if (!h->nccl) {
    // host transport: the same enqueue-and-return contract.  One call in flight per communicator (the staging buffer and
    // `count` belong to it): callers serialise their all-reduces by stream order / events, as they must for RCCL.
    if (count > kHostStaging) { set_error("prost_hip_allreduce_sum_f64: host transport moves at most 64 values"); return 1; }
    hipStream_t s = as_stream(stream);
    h->count = count;
    PH_CHECK(hipMemcpyAsync(h->staging, buf, count * sizeof(double), hipMemcpyDeviceToHost, s));
    PH_CHECK(hipLaunchHostFunc(s, host_allreduce_trampoline, h));
    PH_CHECK(hipMemcpyAsync(buf, h->staging, count * sizeof(double), hipMemcpyHostToDevice, s));
    return 0;
  }
  ncclResult_t r = ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, h->nccl, as_stream(stream));
  if (r != ncclSuccess) return nccl_fail(r, "ncclAllReduce");
  return 0;
}

}  // extern "C"
