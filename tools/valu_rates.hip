// valu_rates.hip -- issue cost of the VALU instruction kinds the fused PDHG kernels are made of, measured on the device:
// cycles per wave64 instruction on one SIMD with 1 and with 3 resident waves (s_memtime around an unrolled block of
// independent instructions).  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_rates tools/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(INSTR)                                                                             \
  for (int it = 0; it < iters; it++) {                                                          \
    asm volatile(REP8(INSTR) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b0), "v"(e0)); \
  }

template <int K>
__global__ void __launch_bounds__(64) rate_kernel(unsigned long long* out, float* sink, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = 1.0001f;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, e0 = 1.0000001;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (K == 0) BODY("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n")
  if (K == 1) BODY("v_pk_fma_f32 %4, %4, %9, %9\n v_pk_fma_f32 %5, %5, %9, %9\n v_pk_fma_f32 %6, %6, %9, %9\n v_pk_fma_f32 %7, %7, %9, %9\n")
  if (K == 2) BODY("v_fma_f64 %4, %4, %9, %9\n v_fma_f64 %5, %5, %9, %9\n v_fma_f64 %6, %6, %9, %9\n v_fma_f64 %7, %7, %9, %9\n")
  if (K == 3) BODY("v_mul_f64 %4, %4, %9\n v_mul_f64 %5, %5, %9\n v_mul_f64 %6, %6, %9\n v_mul_f64 %7, %7, %9\n")
  if (K == 4) BODY("v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3\n")
  if (K == 5) BODY("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n")
  if (K == 6) BODY("v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n")
  if (K == 7) BODY("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
  if (K == 8) BODY("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n")
  if (K == 9) BODY("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %8\n")
  if (K == 10) BODY("v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %9\n")
  if (K == 11) BODY("v_pk_mov_b32 %4, %5, %6\n v_pk_mov_b32 %5, %6, %7\n v_pk_mov_b32 %6, %7, %9\n v_pk_mov_b32 %7, %9, %4\n")
  if (K == 12) BODY("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n")
  if (K == 13) BODY("v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9\n")
  if (K == 14) BODY("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n")
  if (K == 15) BODY("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n")
  if (K == 16) BODY("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_sub_u32 %3, %3, %0\n")
  if (K == 17) BODY("v_min3_f32 %0, %0, %1, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_med3_f32 %3, %3, %0, %8\n")
  if (K == 18) BODY("v_fma_f32 %0, %0, %8, %8\n v_fma_f64 %4, %4, %9, %9\n v_fma_f32 %1, %1, %8, %8\n v_fma_f64 %5, %5, %9, %9\n")
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3);
}

template <int K>
static void run(const char* name, int instr_per_rep) {
  const int iters = 2000;
  for (int waves_per_simd : {1, 2, 3}) {
    const int blocks = 256 * 4 * waves_per_simd;
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * sizeof(unsigned long long)); hipMalloc(&sink, blocks * 64 * sizeof(float));
    hipLaunchKernelGGL(rate_kernel<K>, dim3(blocks), dim3(64), 0, 0, out, sink, iters);
    hipLaunchKernelGGL(rate_kernel<K>, dim3(blocks), dim3(64), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), out, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double per_instr_wave = mean / ((double)iters * 8 * instr_per_rep);
    printf("%-34s %d waves/SIMD: %6.2f cycles per instruction per wave -> %5.2f cycles per instruction per SIMD\n", name, waves_per_simd, per_instr_wave,
           per_instr_wave / waves_per_simd);
    hipFree(out); hipFree(sink);
  }
}

int main() {
  run<0>("v_fma_f32", 4); run<12>("v_add_f32", 4); run<1>("v_pk_fma_f32", 4); run<13>("v_pk_add_f32", 4); run<2>("v_fma_f64", 4); run<3>("v_mul_f64", 4);
  run<18>("v_fma_f32 + v_fma_f64 alternating", 4);
  run<4>("v_cvt_f64_f32", 4); run<5>("v_cvt_f32_f64", 4); run<6>("v_rcp_f64", 4); run<7>("v_rcp_f32", 4); run<8>("v_sqrt_f32", 4);
  run<9>("v_mov_b32", 4); run<10>("v_mov_b64", 4); run<11>("v_pk_mov_b32", 4); run<14>("v_cmp + v_cndmask", 4); run<15>("v_mov_b32_dpp wave_shr/shl", 4);
  run<16>("v_add/and/sub_u32", 4); run<17>("v_min3/max/min/med3_f32", 4);
  return 0;
}
