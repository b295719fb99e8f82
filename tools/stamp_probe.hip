// tools/stamp_probe.hip -- what does it cost to time EVERY kernel of a back-to-back chain with HIP events?
// A chain of N launches of a ~100 us streaming kernel (the size of the headline kernel) is run
//   plain : hipLaunchKernelGGL, one hipEventRecord before and one after the chain
//   pair  : hipExtLaunchKernelGGL(start, stop) -- the start event is a marker of its own in front of the kernel
//   stop  : hipExtLaunchKernelGGL(nullptr, stop) -- only the kernel's own command is stamped; hipEventElapsedTime(stop, stop) is the
//           command's end - start
// each with default events and with hipEventDisableSystemFence events (no system-scope release at the end of the stamped kernel).
// Prints the chain time per launch and the mean of the per-launch samples.
//   hipcc --offload-arch=gfx950 -O2 tools/stamp_probe.hip -o tools/bin/stamp_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stream_kernel(f4* __restrict__ out, const f4* __restrict__ in, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { f4 v = in[i]; v.x *= 1.0001f; v.y += 1.0f; v.z -= 1.0f; v.w *= 0.9999f; __builtin_nontemporal_store(v, out + i); }
}

int main() {
  const size_t n4 = (size_t)14 * 4096 * 4096 / 4 / 4;       // 3.5 floats/pixel in + 3.5 out at 4096^2: the headline kernel's compulsory bytes
  f4 *a, *b;
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16));
  CK(hipMemset(a, 0, n4 * 16));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int N = 200;
  const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int flags = 0; flags < 2; flags++) {
    std::vector<hipEvent_t> es(N), ee(N);
    for (int i = 0; i < N; i++) {
      CK(hipEventCreateWithFlags(&es[i], flags ? hipEventDisableSystemFence : hipEventDefault));
      CK(hipEventCreateWithFlags(&ee[i], flags ? hipEventDisableSystemFence : hipEventDefault));
    }
    for (int mode = 0; mode < 3; mode++) {
      for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(stream_kernel, grid, block, 0, s, (i & 1) ? a : b, (i & 1) ? b : a, n4);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(t0, s));
        for (int i = 0; i < N; i++) {
          f4* o = (i & 1) ? a : b; const f4* in = (i & 1) ? b : a;
          if (mode == 0) hipLaunchKernelGGL(stream_kernel, grid, block, 0, s, o, in, n4);
          else if (mode == 1) hipExtLaunchKernelGGL(stream_kernel, grid, block, 0, s, es[i], ee[i], 0, o, in, n4);
          else hipExtLaunchKernelGGL(stream_kernel, grid, block, 0, s, nullptr, ee[i], 0, o, in, n4);
        }
        CK(hipEventRecord(t1, s));
        CK(hipEventSynchronize(t1));
        float chain = 0; CK(hipEventElapsedTime(&chain, t0, t1));
        double sum = 0; int cnt = 0;
        if (mode > 0) for (int i = 0; i < N; i++) {
          float ms = 0; hipError_t e = hipEventElapsedTime(&ms, mode == 1 ? es[i] : ee[i], ee[i]);
          if (e == hipSuccess) { sum += ms; cnt++; }
        }
        printf("%-22s %-5s chain %.2f us/launch   samples %d mean %.2f us\n", flags ? "DisableSystemFence" : "default events", mode == 0 ? "plain" : mode == 1 ? "pair" : "stop",
               1e3 * chain / N, cnt, cnt ? 1e3 * sum / cnt : 0.0);
      }
    }
    for (int i = 0; i < N; i++) { (void)hipEventDestroy(es[i]); (void)hipEventDestroy(ee[i]); }
  }
  return 0;
}
