// stream_probe.hip -- what does HBM3E on one MI355X actually deliver for multi-stream elementwise
// kernels?  Calibrates the generic kernels' skeleton (elementwise.hpp): bytes per lane, grid size,
// grid-stride vs contiguous chunk per workgroup, non-temporal stores, operand base skew.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_probe tools/stream_probe.hip && gpurun_out/stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NIN> struct In { const float* p[NIN]; };

// MODE 0: grid-stride; MODE 1: each workgroup owns one contiguous chunk
template <int VEC, int NIN, bool NT, int MODE>
__global__ void __launch_bounds__(256) probe(float* out, In<NIN> in, size_t n) {
  const size_t nv = n / VEC;
  size_t i, end, step;
  if (MODE == 0) { i = (size_t)blockIdx.x * 256 + threadIdx.x; end = nv; step = (size_t)gridDim.x * 256; }
  else {
    const size_t per = ((nv + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    i = (size_t)blockIdx.x * per + threadIdx.x; end = i - threadIdx.x + per; if (end > nv) end = nv; step = 256;
  }
  float acc = 0;
  for (; i < end; i += step) {
    if (VEC == 4) {
      f4 v[NIN];
#pragma unroll
      for (int k = 0; k < NIN; k++) v[k] = reinterpret_cast<const f4*>(in.p[k])[i];
      f4 o = v[0];
#pragma unroll
      for (int k = 1; k < NIN; k++) o = o * 0.5f + v[k];
      if (out) { if (NT) __builtin_nontemporal_store(o, reinterpret_cast<f4*>(out) + i); else reinterpret_cast<f4*>(out)[i] = o; }
      else acc += o.x + o.y + o.z + o.w;
    } else {
      float v[NIN];
#pragma unroll
      for (int k = 0; k < NIN; k++) v[k] = in.p[k][i];
      float o = v[0];
#pragma unroll
      for (int k = 1; k < NIN; k++) o = o * 0.5f + v[k];
      if (out) { if (NT) __builtin_nontemporal_store(o, out + i); else out[i] = o; }
      else acc += o;
    }
  }
  if (!out && acc == 123.456f) printf("x");
}

static hipEvent_t e0, e1;
template <class L>
static float time_ms(L launch, int reps = 20) {
  for (int i = 0; i < 3; i++) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int VEC, int NIN, bool NT, int MODE>
static void run(const char* tag, float* out, float** bufs, size_t n, unsigned grid, bool write) {
  In<NIN> in;
  for (int k = 0; k < NIN; k++) in.p[k] = bufs[k];
  float ms = time_ms([&] { hipLaunchKernelGGL((probe<VEC, NIN, NT, MODE>), dim3(grid), dim3(256), 0, 0, write ? out : nullptr, in, n); });
  double bytes = (double)(NIN + (write ? 1 : 0)) * n * 4;
  printf("{\"probe\": \"%s\", \"reads\": %d, \"writes\": %d, \"vec\": %d, \"nt_store\": %d, \"mode\": \"%s\", \"grid\": %u, \"ms\": %.4f, \"GBps\": %.0f}\n", tag, NIN,
         write ? 1 : 0, VEC, NT ? 1 : 0, MODE ? "chunk" : "stride", grid, ms, bytes / 1e9 / (ms * 1e-3));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const size_t n = (size_t)8192 * 8192;          // 268 MB per operand
  const size_t skew = argc > 1 ? (size_t)atol(argv[1]) : 0;   // floats of base skew between operands
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* pool; const size_t slot = n + (1 << 20);
  CK(hipMalloc(&pool, 7 * slot * 4));
  CK(hipMemset(pool, 0, 7 * slot * 4));
  float* bufs[6]; for (int k = 0; k < 6; k++) bufs[k] = pool + (k + 1) * slot + k * skew;
  float* out = pool;
  printf("{\"skew_floats\": %zu}\n", skew);
  { float ms = time_ms([&] { CK(hipMemcpyAsync(out, bufs[0], n * 4, hipMemcpyDeviceToDevice, 0)); });
    printf("{\"probe\": \"hipMemcpyAsync D2D\", \"ms\": %.4f, \"GBps\": %.0f}\n", ms, 2.0 * n * 4 / 1e9 / (ms * 1e-3)); }
  for (unsigned grid : {1024u, 2048u, 4096u, 8192u, 16384u, 65536u}) {
    run<4, 1, false, 0>("read", out, bufs, n, grid, false);
    run<1, 1, false, 0>("read", out, bufs, n, grid, false);
    run<4, 1, false, 0>("copy", out, bufs, n, grid, true);
    run<4, 1, true, 0>("copy", out, bufs, n, grid, true);
    run<1, 1, false, 0>("copy", out, bufs, n, grid, true);
    run<4, 1, false, 1>("copy", out, bufs, n, grid, true);
    run<4, 4, false, 0>("4r1w", out, bufs, n, grid, true);
    run<4, 4, true, 0>("4r1w", out, bufs, n, grid, true);
    run<1, 4, false, 0>("4r1w", out, bufs, n, grid, true);
    run<1, 4, true, 0>("4r1w", out, bufs, n, grid, true);
    run<4, 4, false, 1>("4r1w", out, bufs, n, grid, true);
    run<4, 4, true, 1>("4r1w", out, bufs, n, grid, true);
    run<4, 2, false, 0>("2r1w", out, bufs, n, grid, true);
    run<1, 2, false, 0>("2r1w", out, bufs, n, grid, true);
  }
  return 0;
}
