// Micro-benchmarks (not product code): VALU issue rates of the fp16 ops and LDS read rates on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define VALU_KERNEL(NAME, ASM)                                                     \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters, unsigned seed) { \
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15; \
    unsigned w = 0x3c003c00u;                                                      \
    for (int i = 0; i < iters; ++i) {                                              \
      REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));) \
    }                                                                              \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;   \
  }
// each asm = 8 independent instructions
VALU_KERNEL(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %8\n v_pk_mul_f16 %1, %1, %8\n v_pk_mul_f16 %2, %2, %8\n v_pk_mul_f16 %3, %3, %8\n v_pk_mul_f16 %4, %4, %8\n v_pk_mul_f16 %5, %5, %8\n v_pk_mul_f16 %6, %6, %8\n v_pk_mul_f16 %7, %7, %8")
VALU_KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_add_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n v_pk_add_f16 %4, %4, %8\n v_pk_add_f16 %5, %5, %8\n v_pk_add_f16 %6, %6, %8\n v_pk_add_f16 %7, %7, %8")
VALU_KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %8, %0\n v_pk_fma_f16 %1, %1, %8, %1\n v_pk_fma_f16 %2, %2, %8, %2\n v_pk_fma_f16 %3, %3, %8, %3\n v_pk_fma_f16 %4, %4, %8, %4\n v_pk_fma_f16 %5, %5, %8, %5\n v_pk_fma_f16 %6, %6, %8, %6\n v_pk_fma_f16 %7, %7, %8, %7")
VALU_KERNEL(k_mul_f16, "v_mul_f16 %0, %0, %8\n v_mul_f16 %1, %1, %8\n v_mul_f16 %2, %2, %8\n v_mul_f16 %3, %3, %8\n v_mul_f16 %4, %4, %8\n v_mul_f16 %5, %5, %8\n v_mul_f16 %6, %6, %8\n v_mul_f16 %7, %7, %8")
VALU_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7")
VALU_KERNEL(k_add_u32, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8")
VALU_KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %8, %8, %0 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %1, %8, %8, %1 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %2, %8, %8, %2 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %3, %8, %8, %3 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %4, %8, %8, %4 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %5, %8, %8, %5 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %6, %8, %8, %6 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %7, %8, %8, %7 op_sel_hi:[1,1,0]")
VALU_KERNEL(k_dot2_f32_f16, "v_dot2_f32_f16 %0, %8, %8, %0\n v_dot2_f32_f16 %1, %8, %8, %1\n v_dot2_f32_f16 %2, %8, %8, %2\n v_dot2_f32_f16 %3, %8, %8, %3\n v_dot2_f32_f16 %4, %8, %8, %4\n v_dot2_f32_f16 %5, %8, %8, %5\n v_dot2_f32_f16 %6, %8, %8, %6\n v_dot2_f32_f16 %7, %8, %8, %7")
VALU_KERNEL(k_dot2c_f32_f16, "v_dot2c_f32_f16 %0, %8, %8\n v_dot2c_f32_f16 %1, %8, %8\n v_dot2c_f32_f16 %2, %8, %8\n v_dot2c_f32_f16 %3, %8, %8\n v_dot2c_f32_f16 %4, %8, %8\n v_dot2c_f32_f16 %5, %8, %8\n v_dot2c_f32_f16 %6, %8, %8\n v_dot2c_f32_f16 %7, %8, %8")

// LDS read throughput: every wave reads 8 x ds_read_b64 per inner step at byte misalignment MIS
template <int MIS, int WIDTH>
__global__ __launch_bounds__(256) void k_lds(unsigned *out, int iters) {
  extern __shared__ unsigned char lds[];
  for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((unsigned *)lds)[i] = i;
  __syncthreads();
  unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds + (threadIdx.x & 63) * WIDTH + MIS + (threadIdx.x >> 6) * 4096;
  unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0, s = 0;
  for (int i = 0; i < iters; ++i) {
    if (WIDTH == 8)
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:768\n ds_read_b64 %2, %4 offset:1536\n ds_read_b64 %3, %4 offset:2304\n s_waitcnt lgkmcnt(0)"
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr));
    else {
      unsigned b0, b1, b2, b3;
      asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:768\n ds_read_b32 %2, %4 offset:1536\n ds_read_b32 %3, %4 offset:2304\n s_waitcnt lgkmcnt(0)"
                   : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
      a0 = b0; a1 = b1; a2 = b2; a3 = b3;
    }
    s += a0 ^ a1 ^ a2 ^ a3;
  }
  out[blockIdx.x * 256 + threadIdx.x] = (unsigned)s;
}

template <typename F> float time_ms(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  unsigned *out; CHECK(hipMalloc(&out, 256 * 4096 * 4));
  const int blocks = 256 * 8, iters = 2000;   // 8 blocks/CU = 8 waves/SIMD
  const double insts_per_simd = (double)blocks * 4 / (256 * 4) * iters * 64;  // wave-instrs per SIMD
#define RUNV(K) { float ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, out, iters, 1u); }); \
    printf("%-16s %8.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", #K, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4); }
  RUNV(k_add_u32) RUNV(k_fma_f32) RUNV(k_mul_f16) RUNV(k_pk_mul_f16) RUNV(k_pk_add_f16) RUNV(k_pk_fma_f16) RUNV(k_fma_mix) RUNV(k_dot2_f32_f16) RUNV(k_dot2c_f32_f16)
  CHECK(hipFuncSetAttribute((const void *)k_lds<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 40000));
  const int lblocks = 256 * 4, liters = 4000;
  const double bytes_per_cu8 = (double)lblocks / 256 * 4 * liters * 4 * 64 * 8;
#define RUNL(M, W) { float ms = time_ms([&] { hipLaunchKernelGGL((k_lds<M, W>), dim3(lblocks), dim3(256), 36000, 0, out, liters); }); \
    printf("ds_read_b%d mis=%d  %8.3f ms -> %.1f B/clk/CU @2.4GHz\n", W * 8, M, ms, bytes_per_cu8 * W / 8 / (ms * 1e-3 * 2.4e9)); }
  RUNL(0, 8) RUNL(2, 8) RUNL(4, 8) RUNL(6, 8) RUNL(0, 4) RUNL(2, 4)
  return 0;
}
