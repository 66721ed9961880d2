// Micro-benchmark (not product code): how many cycles a CU's vector-memory path needs per 64-lane buffer load, by
// access width, when every wave of the CU streams loads (data L2-resident, consecutive lanes on consecutive elements,
// rows 2 bytes off dword alignment like the blur's odd-width images).  Loads are inline asm so nothing is merged.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define LD16(OP, DST) \
  asm volatile(OP " %0, %8, %9, %10 offen\n\t" OP " %1, %8, %9, %11 offen\n\t" OP " %2, %8, %9, %12 offen\n\t" OP " %3, %8, %9, %13 offen\n\t" \
               OP " %4, %8, %9, %10 offen offset:2048\n\t" OP " %5, %8, %9, %11 offen offset:2048\n\t" OP " %6, %8, %9, %12 offen offset:2048\n\t" OP " %7, %8, %9, %13 offen offset:2048\n\t" \
               : "=v"(DST[0]), "=v"(DST[1]), "=v"(DST[2]), "=v"(DST[3]), "=v"(DST[4]), "=v"(DST[5]), "=v"(DST[6]), "=v"(DST[7]) \
               : "v"(voff), "s"(r), "s"(s0), "s"(s1), "s"(s2), "s"(s3))
template <int MODE> __global__ __launch_bounds__(256) void k(const void *buf, unsigned long long *out, int iters, int misalign) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, 1 << 26, 0x00020000);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int BYTES = MODE == 0 ? 2 : (MODE == 1 ? 4 : (MODE == 2 ? 8 : 16));
  unsigned voff = lane * BYTES + misalign + (blockIdx.x & 255) * 65536 + wave * 16384;
  int s0 = 0, s1 = 2666, s2 = 5332, s3 = 7998;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  unsigned acc = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) { unsigned d[8]; LD16("buffer_load_ushort", d); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += d[0] ^ d[7]; }
    if constexpr (MODE == 1) { unsigned d[8]; LD16("buffer_load_dword", d); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += d[0] ^ d[7]; }
    if constexpr (MODE == 2) { u2 d[8]; LD16("buffer_load_dwordx2", d); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += d[0].x ^ d[7].y; }
    if constexpr (MODE == 3) { u4 d[8]; LD16("buffer_load_dwordx4", d); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += d[0].x ^ d[7].w; }
    s0 ^= 64; s2 ^= 64;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = c1 - c0; out[(blockIdx.x * 4 + wave) * 2 + 1] = acc; }
}
int main() {
  void *buf; CHECK(hipMalloc(&buf, 1 << 26)); CHECK(hipMemset(buf, 1, 1 << 26));
  unsigned long long *out; CHECK(hipMalloc(&out, 256 * 8 * 4 * 2 * 8));
  std::vector<unsigned long long> h(256 * 8 * 4 * 2);
  const int iters = 200;
  const char *names[4] = {"ushort  (2 B/lane)", "dword   (4 B/lane)", "dwordx2 (8 B/lane)", "dwordx4 (16 B/lane)"};
#define RUN(M, WPS, MIS) { const int blocks = 256 * (WPS); \
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<M>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, MIS); \
    CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h.data(), out, blocks * 4 * 2 * 8, hipMemcpyDeviceToHost)); \
    std::vector<double> cyc; for (int w = 0; w < blocks * 4; ++w) cyc.push_back((double)h[w * 2]); \
    std::sort(cyc.begin(), cyc.end()); const double n = iters * 8.0; \
    printf("load %s misalign %d, %d waves/SIMD: %6.1f cycles per load per wave -> %5.1f cycles of the CU per load instruction\n", \
           names[M], MIS, WPS, cyc[cyc.size() / 2] / n, cyc[cyc.size() / 2] / n / (4 * (WPS))); }
  RUN(0, 4, 0) RUN(0, 4, 2) RUN(1, 4, 0) RUN(1, 4, 2) RUN(2, 4, 0) RUN(2, 4, 2) RUN(3, 4, 0) RUN(3, 4, 2)
  RUN(0, 8, 2) RUN(1, 8, 2) RUN(2, 8, 2) RUN(3, 8, 2)
  return 0;
}
