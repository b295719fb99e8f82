// common.hpp -- shared plumbing of the gfx950 kernel library (libprost_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>

#include "prost_hip.h"

namespace prost_hip {

void set_error(const std::string& msg);
int fail(hipError_t e, const char* what);

#define PH_CHECK(call)                                              \
  do {                                                              \
    hipError_t e_ = (call);                                         \
    if (e_ != hipSuccess) return ::prost_hip::fail(e_, #call);      \
  } while (0)

// launch epilogue: surfaces launch-configuration errors without synchronising
#define PH_LAUNCH_END(name)                                         \
  do {                                                              \
    hipError_t e_ = hipGetLastError();                              \
    if (e_ != hipSuccess) return ::prost_hip::fail(e_, name);       \
    return 0;                                                       \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves, one per SIMD
// Streaming kernels are launched with one 16-byte group per lane whenever the problem allows it: on
// MI355X a 4-read/1-write stream kernel measures 5.6-5.8 TB/s that way against 4.8-5.0 TB/s with a
// 4096-workgroup grid-stride loop (tools/stream_probe.hip, profiles/r01_stream_probe.txt).
constexpr int kMaxGridStride = 1 << 20;
constexpr int kReduceBlocks = 8192;        // partial slots in the reduction workspace (pairs of doubles)

// Kernel timing (prost_hip_next_launch_events): the next PH_LAUNCH of the calling thread hands the events to hipExtLaunchKernelGGL.
// The STOP event is bound to the kernel's own command: it costs nothing and carries the kernel's end.  A START event is a marker of
// its own in front of the kernel -- a barrier packet that breaks the back-to-back dispatch of consecutive launches: a chain of 200
// launches of a 75 us kernel runs 75.0 us per launch plain or with stop events only, 79.8 us with start + stop pairs (78.7 with
// hipEventDisableSystemFence events; tools/stamp_probe.hip, profiles/r04_stamp_probe.txt).  start -> stop is the kernel's duration;
// the distance between the stop events of consecutive launches is the launch period (it contains the idle time between dependent
// launches).  (hipEventRecord brackets cost two such packets per launch and measure the dispatch gap along with the kernel: a 22 us
// kernel read 14 % long.)
extern thread_local hipEvent_t g_launch_ev_start, g_launch_ev_stop;
// device-resident step-size record the generic PDHG kernels of the calling thread read their step sizes from (prost_hip_use_step_record)
extern thread_local void* g_step_record;
}  // namespace prost_hip
#include <hip/hip_ext.h>
#define PH_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                                        \
  do {                                                                                                                              \
    if (::prost_hip::g_launch_ev_stop) {                                                                                            \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ::prost_hip::g_launch_ev_start, ::prost_hip::g_launch_ev_stop, 0,   \
                            __VA_ARGS__);                                                                                           \
      ::prost_hip::g_launch_ev_start = nullptr; ::prost_hip::g_launch_ev_stop = nullptr;                                            \
    } else {                                                                                                                        \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                                          \
    }                                                                                                                               \
  } while (0)
namespace prost_hip {

inline unsigned grid_for(size_t n, int per_thread = 1) {
  size_t b = (n + (size_t)kBlock * per_thread - 1) / ((size_t)kBlock * per_thread);
  if (b < 1) b = 1;
  if (b > (size_t)kMaxGridStride) b = kMaxGridStride;
  return (unsigned)b;
}

}  // namespace prost_hip
