// Micro-benchmark: vector-memory load issue rate per CU by access width (L1/L2-resident footprint).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int BYTES>
__global__ __launch_bounds__(256) void k_ld(const unsigned char *src, unsigned *out, int iters, int row_bytes) {
  const unsigned char *p = src + (size_t)(blockIdx.x % 64) * 65536 + (threadIdx.x >> 6) * 8192 + (threadIdx.x & 63) * BYTES;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    unsigned v0, v1, v2, v3, v4, v5, v6, v7;
    const unsigned char *q = p + (i & 7) * row_bytes;
    if (BYTES == 2) {
      asm volatile("global_load_ushort %0, %8, off\n global_load_ushort %1, %8, off offset:128\n global_load_ushort %2, %8, off offset:256\n global_load_ushort %3, %8, off offset:384\n"
                   "global_load_ushort %4, %8, off offset:512\n global_load_ushort %5, %8, off offset:640\n global_load_ushort %6, %8, off offset:768\n global_load_ushort %7, %8, off offset:896\n s_waitcnt vmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(q));
    } else if (BYTES == 4) {
      asm volatile("global_load_dword %0, %8, off\n global_load_dword %1, %8, off offset:256\n global_load_dword %2, %8, off offset:512\n global_load_dword %3, %8, off offset:768\n"
                   "global_load_dword %4, %8, off offset:1024\n global_load_dword %5, %8, off offset:1280\n global_load_dword %6, %8, off offset:1536\n global_load_dword %7, %8, off offset:1792\n s_waitcnt vmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(q));
    } else {  // 2-byte-misaligned dword loads
      asm volatile("global_load_dword %0, %8, off offset:2\n global_load_dword %1, %8, off offset:258\n global_load_dword %2, %8, off offset:514\n global_load_dword %3, %8, off offset:770\n"
                   "global_load_dword %4, %8, off offset:1026\n global_load_dword %5, %8, off offset:1282\n global_load_dword %6, %8, off offset:1538\n global_load_dword %7, %8, off offset:1794\n s_waitcnt vmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(q));
    }
    acc += v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename F> float time_ms(F launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  unsigned char *src; unsigned *out;
  (void)hipMalloc(&src, 64 * 65536 + 65536); (void)hipMemset(src, 1, 64 * 65536 + 65536); (void)hipMalloc(&out, 4096 * 256 * 4);
  const int blocks = 256 * 2, iters = 2000;
  for (int wpc = 2; wpc <= 8; wpc *= 2) {
    int b = 256 * wpc;
    double loads_per_cu = (double)wpc * 4 * iters * 8;
#define RUN(B, name) { float ms = time_ms([&] { hipLaunchKernelGGL((k_ld<B>), dim3(b), dim3(256), 0, 0, src, out, iters, 1024); }); \
    printf("%d WG/CU %-22s %7.3f ms -> %.1f ns per wave-load per CU (%.1f cyc @2.2GHz)\n", wpc, name, ms, ms * 1e6 / loads_per_cu, ms * 1e6 / loads_per_cu * 2.2); }
    RUN(2, "ushort (128B/instr)") RUN(4, "dword (256B/instr)") RUN(6, "dword misaligned+2")
  }
  (void)blocks;
  return 0;
}
