// Micro-benchmark (not product code): what one tap of the blur's inner loop costs a SIMD, by the form of its LDS reads.
// 2,048 workgroups of 4 waves (8 waves per SIMD), each wave runs TAPS taps: 16 packed fp16 instructions (8 multiplies + 8 adds,
// as the bit-exact contract demands) on the data the PREVIOUS tap's LDS reads fetched (one tap of look-ahead, two register
// buffers, one s_waitcnt per tap -- the shipped loop's structure), plus per tap:
//   none     no LDS read (the vector ALU alone)
//   4xb64    four ds_read_b64 (shipped: rows i .. i+3 of the lane, 8 bytes each)
//   2xr2b64  two ds_read2_b64 (the same 32 bytes per lane in two instructions)
//   4xb32    four ds_read_b32 (half the bytes)
//   8xb64    eight ds_read_b64 (twice the bytes)
//   k_arith2 / k_arith6 / k_wait2 / k_sc3_2 / k_sc3_6 / k_vadd2: no LDS reads; the sixteen packed instructions alone with 2 or 6
//   taps per loop branch, + one s_waitcnt per tap, + the loop's scalar ALU work per tap, + one plain vector instruction per tap
// Prints shader cycles per wave-tap on a SIMD = launch time x clock / (TAPS x 8), the clock read inside the loop
// (s_memtime over s_memrealtime, median over the waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define ARITH(b0, b1, b2, b3, b4, b5, b6, b7) \
  "v_pk_mul_f16 " b0 ", %8, " b0 " op_sel_hi:[0,1]\n\tv_pk_mul_f16 " b1 ", %8, " b1 " op_sel_hi:[0,1]\n\tv_pk_mul_f16 " b2 ", %8, " b2 " op_sel_hi:[0,1]\n\tv_pk_mul_f16 " b3 ", %8, " b3 " op_sel_hi:[0,1]\n\t" \
  "v_pk_mul_f16 " b4 ", %8, " b4 " op_sel_hi:[0,1]\n\tv_pk_mul_f16 " b5 ", %8, " b5 " op_sel_hi:[0,1]\n\tv_pk_mul_f16 " b6 ", %8, " b6 " op_sel_hi:[0,1]\n\tv_pk_mul_f16 " b7 ", %8, " b7 " op_sel_hi:[0,1]\n\t" \
  "v_pk_add_f16 %0, %0, " b0 "\n\tv_pk_add_f16 %1, %1, " b1 "\n\tv_pk_add_f16 %2, %2, " b2 "\n\tv_pk_add_f16 %3, %3, " b3 "\n\t" \
  "v_pk_add_f16 %4, %4, " b4 "\n\tv_pk_add_f16 %5, %5, " b5 "\n\tv_pk_add_f16 %6, %6, " b6 "\n\tv_pk_add_f16 %7, %7, " b7 "\n\t"
#define ARITH_X ARITH("v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39")
#define ARITH_Y ARITH("v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47")
// reads of one tap into registers base .. base+7 (address register v48 = %9 + wandering offset in s20)
#define RD_NONE(b) ""
#define RD_4B64(b) "v_add_u32 v48, s20, %9\n\tds_read_b64 v[" #b ":" #b "+1], v48\n\tds_read_b64 v[" #b "+2:" #b "+3], v48 offset:448\n\tds_read_b64 v[" #b "+4:" #b "+5], v48 offset:896\n\tds_read_b64 v[" #b "+6:" #b "+7], v48 offset:1344\n\t"
#define RD_2R2(b) "v_add_u32 v48, s20, %9\n\tds_read2_b64 v[" #b ":" #b "+3], v48 offset1:56\n\tds_read2_b64 v[" #b "+4:" #b "+7], v48 offset0:112 offset1:168\n\t"
#define RD_2B128(b) "v_add_u32 v48, s20, %9\n\tds_read_b128 v[" #b ":" #b "+3], v48\n\tds_read_b128 v[" #b "+4:" #b "+7], v48 offset:896\n\t"
#define RD_4B32(b) "v_add_u32 v48, s20, %9\n\tds_read_b32 v[" #b "], v48\n\tds_read_b32 v[" #b "+2], v48 offset:448\n\tds_read_b32 v[" #b "+4], v48 offset:896\n\tds_read_b32 v[" #b "+6], v48 offset:1344\n\t"
#define RD_8B64(b) RD_4B64(b) "ds_read_b64 v[" #b ":" #b "+1], v48 offset:1792\n\tds_read_b64 v[" #b "+2:" #b "+3], v48 offset:2240\n\tds_read_b64 v[" #b "+4:" #b "+5], v48 offset:2688\n\tds_read_b64 v[" #b "+6:" #b "+7], v48 offset:3136\n\t"
#define KERNEL(NAME, RD)                                                                                    \
  __global__ __launch_bounds__(256, 8) void NAME(unsigned *out, int taps) {                                          \
    extern __shared__ unsigned lds[];                                                                                \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                                      \
    for (int i = threadIdx.x; i < 19712 / 4; i += 256) lds[i] = 0x3c003c00u;                                         \
    __syncthreads();                                                                                                 \
    unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;                                         \
    const unsigned w = 0x3c003c00u;                                                                                  \
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds +                          \
                          (unsigned)((wave * 8 + (lane >> 5) * 4) * 448 + (lane & 31) * 8);         \
    unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane(taps / 2);                                               \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();              \
    asm volatile("s_mov_b32 s20, 0\n\t" RD(32) "L" #NAME "%=:\n\t"                                                    \
                 "s_waitcnt lgkmcnt(0)\n\ts_add_u32 s20, s20, 8\n\ts_and_b32 s20, s20, 0xbf\n\t" RD(40) ARITH_X    \
                 "s_waitcnt lgkmcnt(0)\n\ts_add_u32 s20, s20, 8\n\ts_and_b32 s20, s20, 0xbf\n\t" RD(32) ARITH_Y    \
                 "s_sub_u32 %10, %10, 1\n\ts_cmp_lg_u32 %10, 0\n\ts_cbranch_scc1 L" #NAME "%=\n\ts_waitcnt lgkmcnt(0)"       \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                     \
                 : "v"(w), "v"(base), "s"(cnt)                                                                         \
                 : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "s20", "scc", "memory"); \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();              \
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 4] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; out[(blockIdx.x * 4 + wave) * 4 + 1] = (unsigned)(c1 - c0); out[(blockIdx.x * 4 + wave) * 4 + 2] = (unsigned)(r1 - r0); } \
  }
KERNEL(k_none, RD_NONE)
KERNEL(k_4b64, RD_4B64)
KERNEL(k_2r2b64, RD_2R2)
KERNEL(k_4b32, RD_4B32)
KERNEL(k_8b64, RD_8B64)

// scalar-side variants of the loop without LDS reads: what the instructions AROUND the sixteen packed ones cost
#define KERNEL2(NAME, BODY, TAPS_PER_ITER)                                                                           \
  __global__ __launch_bounds__(256, 8) void NAME(unsigned *out, int taps) {                                          \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                                      \
    unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;                                         \
    const unsigned w = 0x3c003c00u, base = 0;                                                                        \
    unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane(taps / TAPS_PER_ITER);                                   \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();              \
    asm volatile("s_mov_b32 s20, 0\n\tv_mov_b32 v32, 0x3c003c00\n\tv_mov_b32 v33, v32\n\tv_mov_b32 v34, v32\n\tv_mov_b32 v35, v32\n\tv_mov_b32 v36, v32\n\tv_mov_b32 v37, v32\n\tv_mov_b32 v38, v32\n\tv_mov_b32 v39, v32\n\t" \
                 "v_mov_b32 v40, v32\n\tv_mov_b32 v41, v32\n\tv_mov_b32 v42, v32\n\tv_mov_b32 v43, v32\n\tv_mov_b32 v44, v32\n\tv_mov_b32 v45, v32\n\tv_mov_b32 v46, v32\n\tv_mov_b32 v47, v32\n\t" \
                 "L" #NAME "%=:\n\t" BODY                                                                              \
                 "s_sub_u32 %10, %10, 1\n\ts_cmp_lg_u32 %10, 0\n\ts_cbranch_scc1 L" #NAME "%=\n\t"                     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                     \
                 : "v"(w), "v"(base), "s"(cnt)                                                                         \
                 : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "s20", "scc", "memory"); \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();              \
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 4] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; out[(blockIdx.x * 4 + wave) * 4 + 1] = (unsigned)(c1 - c0); out[(blockIdx.x * 4 + wave) * 4 + 2] = (unsigned)(r1 - r0); } \
  }
#define SC3 "s_waitcnt lgkmcnt(0)\n\ts_add_u32 s20, s20, 8\n\ts_and_b32 s20, s20, 0xbf\n\t"
#define SC1 "s_waitcnt lgkmcnt(0)\n\t"
KERNEL2(k_arith2, ARITH_X ARITH_Y, 2)                                       // two taps per branch, nothing else
KERNEL2(k_arith6, ARITH_X ARITH_Y ARITH_X ARITH_Y ARITH_X ARITH_Y, 6)       // six taps per branch, nothing else
KERNEL2(k_wait2, SC1 ARITH_X SC1 ARITH_Y, 2)                                // + one s_waitcnt per tap
KERNEL2(k_sc3_2, SC3 ARITH_X SC3 ARITH_Y, 2)                                // + waitcnt + two scalar ALU per tap (= k_none's body)
KERNEL2(k_sc3_6, SC3 ARITH_X SC3 ARITH_Y SC3 ARITH_X SC3 ARITH_Y SC3 ARITH_X SC3 ARITH_Y, 6)
KERNEL2(k_vadd2, "v_add_u32 v48, s20, %9\n\t" ARITH_X "v_add_u32 v48, s20, %9\n\t" ARITH_Y, 2)   // + one plain vector instruction per tap

__global__ __launch_bounds__(256, 8) void k_clk(unsigned long long *o) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - r0 < 2000) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = __builtin_amdgcn_s_memtime() - c0; o[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
int main() {
  unsigned *out; CHECK(hipMalloc(&out, 2048 * 4 * 4 * 4)); std::vector<unsigned> h(2048 * 16);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int taps = 400;
#define RUN(K) { for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(K, dim3(2048), dim3(256), 19712, 0, out, taps); \
    CHECK(hipEventRecord(e0, 0)); for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(K, dim3(2048), dim3(256), 19712, 0, out, taps); \
    CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize()); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20; \
    CHECK(hipMemcpy(h.data(), out, 2048 * 16 * 4, hipMemcpyDeviceToHost)); std::vector<double> clk, life; \
    for (int i = 0; i < 2048 * 4; ++i) { clk.push_back(h[i * 4 + 1] / (h[i * 4 + 2] * 10.0)); life.push_back(h[i * 4 + 2] * 0.01); } \
    std::sort(clk.begin(), clk.end()); std::sort(life.begin(), life.end()); const double ghz = clk[clk.size() / 2]; \
    printf("%-10s %.2f us per launch of %d taps x 8 waves per SIMD = %.1f ns per wave-tap = %.1f cycles at the measured %.2f GHz (wave life p50 %.1f us)\n", #K, ms * 1e3, taps, ms * 1e6 / (taps * 8.0), ms * 1e6 / (taps * 8.0) * ghz, ghz, life[life.size() / 2]); }
  RUN(k_none) RUN(k_4b64) RUN(k_2r2b64) RUN(k_4b32) RUN(k_8b64) RUN(k_none) RUN(k_4b64)
  RUN(k_arith2) RUN(k_arith6) RUN(k_wait2) RUN(k_sc3_2) RUN(k_sc3_6) RUN(k_vadd2) RUN(k_arith2) RUN(k_arith6)
  return 0;
}
