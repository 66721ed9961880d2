// Micro-benchmark (not product code): cost of the blur's window FILL alone on the BASELINE grid (6,600 workgroups of 4 waves,
// 8 per CU), two forms that produce the same LDS window (quad layout: element k = {P[k], P[k+32] | P[k+64], P[k+96]}, 44 rows):
//   A  the shipped form: per row and wave four 2-byte buffer loads per lane (P[lane + 32 m]), 2 v_perm, one ds_write_b64
//   T  contiguous 4-byte loads: per FOUR rows four loads of 256 contiguous bytes (one row each, lane l = dword l), a 4 x 4
//      transpose between registers and 16-lane rows (v_permlane16_swap / v_permlane32_swap), one more load for P[128..151],
//      8 v_perm, two ds_write2_b64 per lane -- 5 loads per 4 rows instead of 16, rows of any 2-byte alignment (the load
//      address is rounded down to 4 bytes and the row lands one element lower in LDS)
//   N  no fill (what the rest of the kernel costs)
// Prints the launch time of each and checks that A and T build the same window.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int PITCH = 57 * 8, ROWS = 44, LDSB = ROWS * PITCH + 16;
typedef unsigned uv2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uv2 lds_uv2;
typedef __attribute__((address_space(3))) unsigned lds_u;

template <int V>
__global__ __launch_bounds__(256, 8) void fill_kernel(const unsigned short *img, int H, int W, int tiles_x, int tiles_y, unsigned long long *out, int cshift) {
  extern __shared__ unsigned lds[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds + 8;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y; b /= tiles_y;       // b = plane index (image * 3 + channel)
  const int c_first = min(max(tx * 128 - 12 + cshift, 2), W - 162), r_first = min(max(ty * 32 - 6, 0), H - 48);
  const unsigned long long pa = (unsigned long long)img + (unsigned long long)b * H * W * 2ull;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)pa, 0, H * W * 2, 0x00020000);
  const int w2 = W * 2;
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
  } else if constexpr (V == 1) {
    // groups of four rows: wave w takes groups 3 w .. 3 w + 2 (11 groups in all)
    const int q = lane >> 4, j = lane & 15;
    const int a0 = r_first * W + c_first;                        // pixel index of the window's P[0] in row 0 (plane base 4-byte aligned here)
    unsigned d[3][5];
    const int pbase = (unsigned)(pa & 2) >> 1;                   // parity of the plane's own base (0 here: kept for the general form)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int gi = wave * 3 + g;
      if (gi < 11) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int a = a0 + (gi * 4 + i) * W;                    // scalar
          const int so = __builtin_amdgcn_readfirstlane(2 * (a - ((a + pbase) & 1)));
          d[g][i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)lane * 4u, so, 0);
        }
        const int aq = a0 + (gi * 4 + q) * W;                     // per lane: this lane's row after the transpose
        const int sq = (aq + pbase) & 1;
        d[g][4] = j < 13 ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)(2 * (aq - sq) + 256 + 4 * j), 0, 0) : 0u;
      }
    }
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
        const int sq = (aq + pbase) & 1;
        // element 2j - s: the low halves; element 2j - s + 1: the high halves
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
  // consume: checksum of elements 0..55 of every row
  unsigned long long sum = 0;
  for (int i = threadIdx.x; i < ROWS * 56; i += 256) {
    const int r = i / 56, k = i - r * 56;
    const uv2 e = *(lds_uv2 *)(size_t)(lds0 + (unsigned)(r * PITCH + k * 8));
    sum += (unsigned long long)e.x * (unsigned)(i + 1) + (unsigned long long)e.y * (unsigned)(2 * i + 7);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) atomicAdd(&out[blockIdx.x], sum);
}

int run(int H, int W) {
  const int planes = 24, tiles_x = (W + 127) / 128, tiles_y = (H + 31) / 32, blocks = planes * tiles_x * tiles_y;
  printf("%d planes of %d x %d: %d workgroups\n", planes, H, W, blocks);
  const size_t n = (size_t)planes * H * W;
  std::vector<unsigned short> h(n);
  unsigned x = 12345; for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(x >> 16); }
  unsigned short *img; CHECK(hipMalloc(&img, n * 2 + 64)); CHECK(hipMemcpy(img, h.data(), n * 2, hipMemcpyHostToDevice));
  unsigned long long *out; CHECK(hipMalloc(&out, blocks * 8 * 3));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<unsigned long long> r[3];
  for (int cshift = 0; cshift < 2; ++cshift) {
    float ms[3];
#define RUN(V) { CHECK(hipMemset(out + V * blocks, 0, blocks * 8)); \
      hipLaunchKernelGGL((fill_kernel<V>), dim3(blocks), dim3(256), LDSB, 0, img, H, W, tiles_x, tiles_y, out + V * blocks, cshift); \
      CHECK(hipDeviceSynchronize()); r[V].resize(blocks); CHECK(hipMemcpy(r[V].data(), out + V * blocks, blocks * 8, hipMemcpyDeviceToHost)); \
      for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((fill_kernel<V>), dim3(blocks), dim3(256), LDSB, 0, img, H, W, tiles_x, tiles_y, out + V * blocks, cshift); \
      CHECK(hipEventRecord(e0, 0)); \
      for (int w = 0; w < 200; ++w) hipLaunchKernelGGL((fill_kernel<V>), dim3(blocks), dim3(256), LDSB, 0, img, H, W, tiles_x, tiles_y, out + V * blocks, cshift); \
      CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms[V], e0, e1)); ms[V] /= 200; }
    RUN(0) RUN(1) RUN(2)
    int bad = 0; for (int i = 0; i < blocks; ++i) bad += r[0][i] != r[1][i];
    printf("column shift %d: A (2-byte loads) %.2f us, T (contiguous 4-byte loads + transpose) %.2f us, N (no fill) %.2f us per launch; windows that differ: %d of %d\n",
           cshift, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3, bad, blocks);
  }
  CHECK(hipFree(img)); CHECK(hipFree(out));
  return 0;
}
int main() {
  if (run(800, 1333)) return 1;
  if (run(480, 640)) return 1;      // one round of workgroups: the native-size regime (launch-latency- and first-fill-bound)
  if (run(375, 500)) return 1;
  return 0;
}
