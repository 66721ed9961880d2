// Micro-benchmark (not product code): cost of launching N 256-thread workgroups that return at once, by grid size,
// dynamic-LDS size and register footprint (back-to-back launches on one stream, one event pair).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int REGS> __global__ __launch_bounds__(256) void k_empty(unsigned *out, int never) {
  extern __shared__ unsigned lds[];
  if (never) {   // keeps REGS registers and the LDS allocation alive without executing anything
    unsigned a[REGS];
    for (int i = 0; i < REGS; ++i) a[i] = out[threadIdx.x + i];
    for (int r = 0; r < never; ++r) for (int i = 0; i < REGS; ++i) a[i] = a[i] * 3 + a[(i + 1) % REGS];
    unsigned s = 0; for (int i = 0; i < REGS; ++i) s += a[i];
    lds[threadIdx.x] = s; __syncthreads(); out[threadIdx.x] = lds[(threadIdx.x + 1) & 255];
  }
}
template <typename F> float time_us(F launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / reps;
}
int main() {
  unsigned *out; CHECK(hipMalloc(&out, 1 << 20));
  CHECK(hipFuncSetAttribute((const void *)k_empty<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void *)k_empty<90>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  for (int lds : {0, 36864}) for (int grid : {256, 1024, 1800, 3600, 7200, 14400}) {
    float a = time_us([&] { hipLaunchKernelGGL(k_empty<8>, dim3(grid), dim3(256), lds, 0, out, 0); }, 200);
    float b = time_us([&] { hipLaunchKernelGGL(k_empty<90>, dim3(grid), dim3(256), lds, 0, out, 0); }, 200);
    printf("lds %6d B  grid %6d WGs x 256: %7.2f us (8 regs)  %7.2f us (~100 regs)\n", lds, grid, a, b);
  }
  return 0;
}
