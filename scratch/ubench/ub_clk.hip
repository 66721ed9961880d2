// Micro-benchmark (not product code): shader cycles per packed-fp16 / fp32 VALU instruction at 1..8 waves per SIMD,
// and the clock the chip holds meanwhile (s_memtime = shader cycles, s_memrealtime = 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP8(x) x x x x x x x x
#define VALU_KERNEL(NAME, ASM)                                                     \
  __global__ __launch_bounds__(256) void NAME(unsigned long long *out, int iters, unsigned seed) { \
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15; \
    unsigned w = 0x3c003c00u;                                                      \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    for (int i = 0; i < iters; ++i) {                                              \
      REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));) \
    }                                                                              \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    if ((threadIdx.x & 63) == 0) { unsigned long long *o = out + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; o[0] = c1 - c0; o[1] = r1 - r0; o[2] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; } \
  }
VALU_KERNEL(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %8\n v_pk_mul_f16 %1, %1, %8\n v_pk_mul_f16 %2, %2, %8\n v_pk_mul_f16 %3, %3, %8\n v_pk_mul_f16 %4, %4, %8\n v_pk_mul_f16 %5, %5, %8\n v_pk_mul_f16 %6, %6, %8\n v_pk_mul_f16 %7, %7, %8")
VALU_KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %8, %0\n v_pk_fma_f16 %1, %1, %8, %1\n v_pk_fma_f16 %2, %2, %8, %2\n v_pk_fma_f16 %3, %3, %8, %3\n v_pk_fma_f16 %4, %4, %8, %4\n v_pk_fma_f16 %5, %5, %8, %5\n v_pk_fma_f16 %6, %6, %8, %6\n v_pk_fma_f16 %7, %7, %8, %7")
VALU_KERNEL(k_mul_f16, "v_mul_f16 %0, %0, %8\n v_mul_f16 %1, %1, %8\n v_mul_f16 %2, %2, %8\n v_mul_f16 %3, %3, %8\n v_mul_f16 %4, %4, %8\n v_mul_f16 %5, %5, %8\n v_mul_f16 %6, %6, %8\n v_mul_f16 %7, %7, %8")
VALU_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7")
VALU_KERNEL(k_add_u32, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8")
VALU_KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_add_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n v_pk_add_f16 %4, %4, %8\n v_pk_add_f16 %5, %5, %8\n v_pk_add_f16 %6, %6, %8\n v_pk_add_f16 %7, %7, %8")
VALU_KERNEL(k_pk_mul_add, "v_pk_mul_f16 %0, %1, %8\n v_pk_add_f16 %2, %2, %0\n v_pk_mul_f16 %3, %4, %8\n v_pk_add_f16 %5, %5, %3\n v_pk_mul_f16 %0, %6, %8\n v_pk_add_f16 %7, %7, %0\n v_pk_mul_f16 %3, %1, %8\n v_pk_add_f16 %2, %2, %3")
VALU_KERNEL(k_pk_plus_plain, "v_pk_mul_f16 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_pk_mul_f16 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_pk_mul_f16 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_pk_mul_f16 %6, %6, %8\n v_add_u32 %7, %7, %8")
VALU_KERNEL(k_add_f16, "v_add_f16 %0, %0, %8\n v_add_f16 %1, %1, %8\n v_add_f16 %2, %2, %8\n v_add_f16 %3, %3, %8\n v_add_f16 %4, %4, %8\n v_add_f16 %5, %5, %8\n v_add_f16 %6, %6, %8\n v_add_f16 %7, %7, %8")
VALU_KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8")

int main() {
  unsigned long long *out; CHECK(hipMalloc(&out, 256 * 8 * 4 * 4 * 8));
  const int iters = 4000;
  hipEvent_t ev0, ev1; CHECK(hipEventCreate(&ev0)); CHECK(hipEventCreate(&ev1));
  std::vector<unsigned long long> h(256 * 8 * 4 * 4);
  // warm the clocks
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_pk_mul_f16, dim3(2048), dim3(256), 0, 0, out, iters, 1u);
  CHECK(hipDeviceSynchronize());
#define RUN(K, WPS) { const int blocks = 256 * (WPS); \
    hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, out, iters, 1u); \
    CHECK(hipEventRecord(ev0, 0)); \
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, out, iters, 1u); \
    CHECK(hipEventRecord(ev1, 0)); \
    CHECK(hipDeviceSynchronize()); float ms = 0; CHECK(hipEventElapsedTime(&ms, ev0, ev1)); ms /= 3; CHECK(hipMemcpy(h.data(), out, blocks * 4 * 4 * 8, hipMemcpyDeviceToHost)); \
    std::vector<double> cyc, clk; for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[w * 4]); clk.push_back((double)h[w * 4] / (double)h[w * 4 + 1] * 0.1); } \
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end()); \
    const double n = (double)iters * 64; \
    const double wall_cyc = ms * 1e-3 * clk[clk.size() / 2] * 1e9; \
    printf("%-14s %d waves/SIMD: %.2f cycles per wave-instr per wave  -> %.2f per SIMD if co-resident; clock %.2f GHz; launch %.3f ms = %.2f cycles per wave-instr per SIMD by wall time (wave life / launch = %.2f)\n", #K, WPS, cyc[cyc.size() / 2] / n, cyc[cyc.size() / 2] / n / (WPS), clk[clk.size() / 2], ms, wall_cyc / (n * (WPS)), cyc[cyc.size() / 2] / wall_cyc); }
  RUN(k_pk_mul_f16, 1) RUN(k_pk_mul_f16, 2) RUN(k_pk_mul_f16, 3) RUN(k_pk_mul_f16, 4) RUN(k_pk_mul_f16, 8)
  RUN(k_pk_fma_f16, 1) RUN(k_pk_fma_f16, 2) RUN(k_pk_fma_f16, 4) RUN(k_pk_fma_f16, 8)
  RUN(k_mul_f16, 1) RUN(k_mul_f16, 2) RUN(k_mul_f16, 4) RUN(k_mul_f16, 8)
  RUN(k_fma_f32, 1) RUN(k_fma_f32, 2) RUN(k_fma_f32, 4) RUN(k_fma_f32, 8)
  RUN(k_add_u32, 1) RUN(k_add_u32, 2) RUN(k_add_u32, 4) RUN(k_add_u32, 8)
  RUN(k_pk_add_f16, 2) RUN(k_pk_add_f16, 8) RUN(k_pk_mul_add, 2) RUN(k_pk_mul_add, 8) RUN(k_pk_plus_plain, 2) RUN(k_pk_plus_plain, 8)
  RUN(k_add_f16, 2) RUN(k_add_f16, 8) RUN(k_pk_add_u16, 2) RUN(k_pk_add_u16, 8)
  return 0;
}
