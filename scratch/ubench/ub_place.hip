// Micro-benchmark (not product code): how the dispatcher spreads ONE round of workgroups (256 threads, 19.7 KB of LDS: up to 8 per
// CU) over the 256 CUs, for a 1-D grid of exactly the working workgroups and for a 2-D grid a third of whose workgroups exit at
// once (the blur's ragged-batch grid).  Prints the histogram of resident workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256, 8) void place(unsigned *out, int work_x, int spin) {
  extern __shared__ unsigned lds[];
  if ((int)blockIdx.x >= work_x) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
    out[blockIdx.y * gridDim.x + blockIdx.x] = 0x80000000u | (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
  }
  lds[threadIdx.x] = 1;
}
int main() {
  unsigned *out; CHECK(hipMalloc(&out, 4096 * 4));
  std::vector<unsigned> h(4096);
  struct { int gx, gy, work_x; const char *name; } cfg[] = {{1635, 1, 1635, "1-D grid, 1,635 working"}, {304, 8, 204, "2-D grid 304 x 8, 204 working per row (1,632)"},
                                                              {1792, 1, 1792, "1-D grid, 1,792 (7 per CU)"}, {2048, 1, 2048, "1-D grid, 2,048 (8 per CU)"}, {1280, 1, 1280, "1-D grid, 1,280 (5 per CU)"}};
  for (auto &c : cfg) {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(out, 0, 4096 * 4));
      hipLaunchKernelGGL(place, dim3(c.gx, c.gy), dim3(256), 19712, 0, out, c.work_x, 1000);
      CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h.data(), out, 4096 * 4, hipMemcpyDeviceToHost));
      std::map<unsigned, int> per; for (unsigned v : h) if (v & 0x80000000u) per[v & 0xfff]++;
      std::map<int, int> hist; for (auto &p : per) hist[p.second]++;
      printf("%s: %zu CUs;", c.name, per.size()); for (auto &p : hist) printf(" %d CUs x %d", p.second, p.first); printf("\n");
    }
  }
  return 0;
}
