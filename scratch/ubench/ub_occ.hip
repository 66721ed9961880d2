// Micro-benchmark (not product code): how many 256-thread workgroups does a CU actually hold at once, by dynamic-LDS size?
// Each workgroup spins for 20 us on the 100 MHz wall clock and records (begin, end, HW_ID, XCC_ID); the host counts the peak
// number of simultaneously resident workgroups per CU and over the chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int REGS> __global__ __launch_bounds__(256) void k_spin(unsigned long long *out, int ticks, int never) {
  extern __shared__ unsigned lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  lds[threadIdx.x] = threadIdx.x;
  if (never) {   // keeps REGS registers allocated without executing anything
    unsigned a[REGS];
    for (int i = 0; i < REGS; ++i) a[i] = (unsigned)out[threadIdx.x + i];
    for (int r = 0; r < never; ++r) for (int i = 0; i < REGS; ++i) a[i] = a[i] * 3 + a[(i + 1) % REGS];
    unsigned s = 0; for (int i = 0; i < REGS; ++i) s += a[i];
    lds[threadIdx.x] = s;
  }
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long *o = out + (size_t)blockIdx.x * 4;
    o[0] = t0; o[1] = __builtin_amdgcn_s_memrealtime();
    o[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20) + lds[1] - 1;
  }
}
template <int REGS> int run(const char *what);
int main() { return run<4>("few registers") || run<44>("~50 registers") || run<58>("~64 registers") || run<66>("~72 registers"); }
template <int REGS> int run(const char *what) {
  const int grid = 6144;
  hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void *)k_spin<REGS>));
  printf("== %s: %d VGPRs\n", what, fa.numRegs);
  unsigned long long *out; CHECK(hipMalloc(&out, grid * 32));
  std::vector<unsigned long long> h(grid * 4);
  CHECK(hipFuncSetAttribute((const void *)k_spin<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  for (int lds : {1024, 16384, 18432, 20480, 23040}) {
    int nb = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spin<REGS>, 256, lds));
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k_spin<REGS>, dim3(grid), dim3(256), lds, 0, out, 2000, 0); }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), out, grid * 32, hipMemcpyDeviceToHost));
    std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
    std::vector<std::pair<unsigned long long, int>> all;
    for (int b = 0; b < grid; ++b) {
      const unsigned hw = (unsigned)h[b * 4 + 2], xcc = (unsigned)h[b * 4 + 3] & 0xf;
      const unsigned cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
      ev[cu].push_back({h[b * 4], +1}); ev[cu].push_back({h[b * 4 + 1], -1});
      all.push_back({h[b * 4], +1}); all.push_back({h[b * 4 + 1], -1});
    }
    auto peak = [](std::vector<std::pair<unsigned long long, int>> &v) { std::sort(v.begin(), v.end()); int c = 0, p = 0; for (auto &e : v) { c += e.second; p = std::max(p, c); } return p; };
    unsigned wmax = 0; for (int b = 0; b < grid; ++b) wmax = std::max(wmax, (unsigned)h[b * 4 + 2] & 0xf);
    std::map<int, int> hist; for (auto &kv : ev) hist[peak(kv.second)]++;
    printf("lds %6d B: runtime says %2d per CU; measured on %zu CUs, chip peak %4d; per-CU peak histogram:", lds, nb, ev.size(), peak(all));
    for (auto &kv : hist) printf(" %d x%d", kv.first, kv.second);
    printf("; highest wave slot %u\n", wmax);
  }
  return 0;
}
