// Micro-benchmark (not product code): how long a workgroup's window FILL takes while older workgroups of the same CU sit in
// their tap loops.  One round of 2,048 workgroups (8 per CU); the first K x 256 ("tappers", dispatched first = older) run a
// tap-loop-like stream (per tap 4 ds_read_b64 + 16 packed fp16 instructions + the loop's scalar work) for the whole launch, the
// others ("fillers") do the shipped fill (44 rows, four 2-byte loads per row and lane) or the contiguous 4-byte form and
// stamp (100 MHz clock) start / loads issued / window ready.  Prints the fillers' medians per K.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int PITCH = 57 * 8, ROWS = 44, LDSB = ROWS * PITCH + 16;
typedef unsigned uv2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uv2 lds_uv2;

template <int V, int PRIO>
__global__ __launch_bounds__(256, 8) void mix_kernel(const unsigned short *img, int H, int W, int tappers, int taps, unsigned long long *out) {
  extern __shared__ unsigned lds[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds + 8;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  if ((int)blockIdx.x < tappers) {
    // tap-loop-like stream: the addresses wander over the window, the data is whatever LDS holds
    unsigned a0 = 0x3c003c00u, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    unsigned addr = lds0 + (unsigned)((wave * 8 + (lane >> 5) * 4) * PITCH + (lane & 31) * 8);
    const unsigned w = 0x3c003c00u;
    for (int t = 0; t < taps; ++t) {
      uv2 x0, x1, x2, x3;
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:456\n\tds_read_b64 %2, %4 offset:912\n\tds_read_b64 %3, %4 offset:1368\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(addr + (unsigned)((t % 12) * PITCH + (t % 24) * 8)) : "memory");
      asm volatile("v_pk_mul_f16 %8, %8, %16\n\tv_pk_mul_f16 %9, %9, %16\n\tv_pk_mul_f16 %10, %10, %16\n\tv_pk_mul_f16 %11, %11, %16\n\t"
                   "v_pk_mul_f16 %12, %12, %16\n\tv_pk_mul_f16 %13, %13, %16\n\tv_pk_mul_f16 %14, %14, %16\n\tv_pk_mul_f16 %15, %15, %16\n\t"
                   "v_pk_add_f16 %0, %0, %8\n\tv_pk_add_f16 %1, %1, %9\n\tv_pk_add_f16 %2, %2, %10\n\tv_pk_add_f16 %3, %3, %11\n\t"
                   "v_pk_add_f16 %4, %4, %12\n\tv_pk_add_f16 %5, %5, %13\n\tv_pk_add_f16 %6, %6, %14\n\tv_pk_add_f16 %7, %7, %15"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                     "+v"(x0.x), "+v"(x0.y), "+v"(x1.x), "+v"(x1.y), "+v"(x2.x), "+v"(x2.y), "+v"(x3.x), "+v"(x3.y) : "v"(w));
    }
    if (lane == 0 && wave == 0) { out[blockIdx.x * 4] = t_start; out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); out[blockIdx.x * 4 + 1] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; }
    return;
  }
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  int b = blockIdx.x % 1800;                       // 24 planes of 480 x 640: 5 x 15 tiles each
  const int tx = b % 5; b /= 5;
  const int ty = b % 15; b /= 15;
  const int c_first = min(max(tx * 128 - 12, 2), W - 162), r_first = min(max(ty * 32 - 6, 0), H - 48);
  const unsigned long long pa = (unsigned long long)img + (unsigned long long)b * H * W * 2ull;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)pa, 0, H * W * 2, 0x00020000);
  const int w2 = W * 2;
  unsigned long long t_issued = 0;
  if constexpr (V == 0) {
    const int qb = wave * 11;
    short v[11][4];
    unsigned coff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) coff[k] = 2u * (unsigned)(c_first + lane) + 64u * k;
    const int s0 = (r_first + qb) * w2;
#pragma unroll
    for (int g = 0; g < 11; ++g) {
      const int so = __builtin_amdgcn_readfirstlane(s0 + g * w2);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[g][k] = __builtin_amdgcn_raw_buffer_load_b16(rsrc, coff[k], so, 0);
    }
    t_issued = __builtin_amdgcn_s_memrealtime();
    if (lane < 56) {
#pragma unroll
      for (int g = 0; g < 11; ++g) {
        typedef short s2v __attribute__((ext_vector_type(2)));
        uv2 e;
        e.x = __builtin_bit_cast(unsigned, s2v{v[g][0], v[g][1]});
        e.y = __builtin_bit_cast(unsigned, s2v{v[g][2], v[g][3]});
        *(lds_uv2 *)(size_t)(lds0 + (unsigned)((qb + g) * PITCH + lane * 8)) = e;
      }
    }
  } else {
    const int q = lane >> 4, j = lane & 15;
    const int a0 = r_first * W + c_first;
    unsigned d[3][5];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int gi = wave * 3 + g;
      if (gi < 11) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int a = a0 + (gi * 4 + i) * W;
          const int so = __builtin_amdgcn_readfirstlane(2 * (a - (a & 1)));
          d[g][i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)lane * 4u, so, 0);
        }
        const int aq = a0 + (gi * 4 + q) * W;
        const int sq = aq & 1;
        d[g][4] = j < 13 ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)(2 * (aq - sq) + 256 + 4 * j), 0, 0) : 0u;
      }
    }
    t_issued = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int gi = wave * 3 + g;
      if (gi < 11) {
        unsigned x0 = d[g][0], x1 = d[g][1], x2 = d[g][2], x3 = d[g][3];
        asm volatile("s_nop 0\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 0\n\t"
                     "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 0"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        const unsigned x4 = d[g][4];
        const int aq = a0 + (gi * 4 + q) * W;
        const int sq = aq & 1;
        uv2 ea, eb, ec, ed;
        ea.x = __builtin_amdgcn_perm(x1, x0, 0x05040100u); ea.y = __builtin_amdgcn_perm(x3, x2, 0x05040100u);
        eb.x = __builtin_amdgcn_perm(x1, x0, 0x07060302u); eb.y = __builtin_amdgcn_perm(x3, x2, 0x07060302u);
        ec.x = __builtin_amdgcn_perm(x2, x1, 0x05040100u); ec.y = __builtin_amdgcn_perm(x4, x3, 0x05040100u);
        ed.x = __builtin_amdgcn_perm(x2, x1, 0x07060302u); ed.y = __builtin_amdgcn_perm(x4, x3, 0x07060302u);
        const unsigned addr = lds0 + (unsigned)((gi * 4 + q) * PITCH + (2 * j - sq) * 8);
        asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" :: "v"(addr), "v"(ea), "v"(eb) : "memory");
        if (j < 12 + sq) asm volatile("ds_write2_b64 %0, %1, %2 offset0:32 offset1:33" :: "v"(addr), "v"(ec), "v"(ed) : "memory");
      }
    }
  }
  __syncthreads();
  const unsigned long long t_ready = __builtin_amdgcn_s_memrealtime();
  const uv2 e = *(lds_uv2 *)(size_t)(lds0 + (unsigned)((threadIdx.x % ROWS) * PITCH + (lane % 56) * 8));
  if (lane == 0 && wave == 0) { out[blockIdx.x * 4] = t_start; out[blockIdx.x * 4 + 1] = t_issued; out[blockIdx.x * 4 + 2] = t_ready; out[blockIdx.x * 4 + 3] = e.x ^ e.y; }
}

int main() {
  const int H = 480, W = 640, planes = 24, blocks = 2048;
  const size_t n = (size_t)planes * H * W;
  unsigned short *img; CHECK(hipMalloc(&img, n * 2 + 64)); CHECK(hipMemset(img, 0x3c, n * 2));
  unsigned long long *out; CHECK(hipMalloc(&out, blocks * 4 * 8));
  std::vector<unsigned long long> h(blocks * 4);
#define RUN(V, P, K, TAPS) { \
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((mix_kernel<V, P>), dim3(blocks), dim3(256), LDSB, 0, img, H, W, (K) * 256, TAPS, out); \
    CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h.data(), out, blocks * 4 * 8, hipMemcpyDeviceToHost)); \
    unsigned long long t0 = ~0ull; for (int i = 0; i < blocks; ++i) t0 = std::min(t0, h[i * 4]); \
    std::vector<double> st, is, rd, te; \
    for (int i = 0; i < (K) * 256; ++i) te.push_back((h[i * 4 + 3] - t0) * 0.01); \
    for (int i = (K) * 256; i < blocks; ++i) { st.push_back((h[i * 4] - t0) * 0.01); is.push_back((h[i * 4 + 1] - h[i * 4]) * 0.01); rd.push_back((h[i * 4 + 2] - h[i * 4]) * 0.01); } \
    std::sort(st.begin(), st.end()); std::sort(is.begin(), is.end()); std::sort(rd.begin(), rd.end()); std::sort(te.begin(), te.end()); \
    printf("%s%s, %d tapper workgroups per CU (%d taps): fillers start p50 %.2f us; start -> loads issued p50 %.2f p90 %.2f us; start -> window ready p50 %.2f p90 %.2f us; tappers end p50 %.2f us\n", \
           V ? "T (4-byte contiguous)" : "A (2-byte)", P ? " prio 3" : "", K, TAPS, st[st.size() / 2], is[is.size() / 2], is[is.size() * 9 / 10], rd[rd.size() / 2], rd[rd.size() * 9 / 10], te.empty() ? 0.0 : te[te.size() / 2]); }
  RUN(0, 0, 0, 0) RUN(0, 0, 1, 150) RUN(0, 0, 2, 150) RUN(0, 0, 4, 100) RUN(0, 0, 6, 60)
  RUN(0, 1, 2, 150) RUN(0, 1, 4, 100) RUN(0, 1, 6, 60)
  RUN(1, 0, 0, 0) RUN(1, 0, 2, 150) RUN(1, 0, 4, 100) RUN(1, 0, 6, 60)
  RUN(1, 1, 4, 100)
  return 0;
}
