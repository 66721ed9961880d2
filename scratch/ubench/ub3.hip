// Micro-benchmark: do scalar / LDS instructions interleaved in a wave's stream cost VALU throughput?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define PK8 "v_pk_mul_f16 %0, %0, %8\n v_pk_mul_f16 %1, %1, %8\n v_pk_mul_f16 %2, %2, %8\n v_pk_mul_f16 %3, %3, %8\n v_pk_mul_f16 %4, %4, %8\n v_pk_mul_f16 %5, %5, %8\n v_pk_mul_f16 %6, %6, %8\n v_pk_mul_f16 %7, %7, %8\n"
#define PKS8 "v_pk_mul_f16 %0, %0, %8\n s_add_u32 %9, %9, 1\n v_pk_mul_f16 %1, %1, %8\n s_add_u32 %9, %9, 1\n v_pk_mul_f16 %2, %2, %8\n s_add_u32 %9, %9, 1\n v_pk_mul_f16 %3, %3, %8\n s_add_u32 %9, %9, 1\n v_pk_mul_f16 %4, %4, %8\n s_add_u32 %9, %9, 1\n v_pk_mul_f16 %5, %5, %8\n s_add_u32 %9, %9, 1\n v_pk_mul_f16 %6, %6, %8\n s_add_u32 %9, %9, 1\n v_pk_mul_f16 %7, %7, %8\n s_add_u32 %9, %9, 1\n"
#define PKL8 "v_pk_mul_f16 %0, %0, %8\n v_pk_mul_f16 %1, %1, %8\n v_pk_mul_f16 %2, %2, %8\n v_pk_mul_f16 %3, %3, %8\n ds_read_b64 %10, %11\n v_pk_mul_f16 %4, %4, %8\n v_pk_mul_f16 %5, %5, %8\n v_pk_mul_f16 %6, %6, %8\n v_pk_mul_f16 %7, %7, %8\n ds_read_b64 %10, %11 offset:768\n s_waitcnt lgkmcnt(1)\n"
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed) {
  __shared__ unsigned long long lds[2048];
  lds[threadIdx.x] = threadIdx.x; __syncthreads();
  unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
  unsigned w = 0x3c003c00u, sc = 0;
  unsigned long long ld = 0;
  unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds + (threadIdx.x & 63) * 8;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) asm volatile(PK8 PK8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
    if (MODE == 1) asm volatile(PKS8 PKS8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(w), "+s"(sc) : : "scc");
    if (MODE == 2) asm volatile(PKL8 PKL8 "s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(w), "+s"(sc), "=&v"(ld) : "v"(addr));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ sc ^ (unsigned)ld;
}
template <typename F> float time_ms(F launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  unsigned *out; (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 2000;
  for (int wpc = 1; wpc <= 8; wpc *= 2) {
    int b = 256 * wpc;
    double pk_per_simd = (double)wpc * iters * 16;
#define RUN(M, name) { float ms = time_ms([&] { hipLaunchKernelGGL((k<M>), dim3(b), dim3(256), 0, 0, out, iters, 1u); }); \
    printf("%d waves/SIMD %-28s %7.3f ms -> %.2f ns per pk op per SIMD\n", wpc, name, ms, ms * 1e6 / pk_per_simd); }
    RUN(0, "pk only") RUN(1, "pk + 1 s_add each") RUN(2, "pk + 1 ds_read_b64 per 4")
  }
  return 0;
}
