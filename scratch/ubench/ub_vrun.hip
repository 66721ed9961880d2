// Micro-benchmark (not product code): the tolerance mode's tap loop with VERTICAL RUNS -- n taps of one PSF column in
// consecutive PSF rows share n + 3 window rows per lane instead of 4 n (the lane's four output rows slide down the column).
// Same frame as ub_tap.hip (2,048 workgroups of 4 waves = 8 waves per SIMD, one group of look-ahead, two register buffers, one
// s_waitcnt per group), arithmetic = ONE v_pk_fma_f16 per register and tap (DIB_ACC_FMA16's):
//   f_none   8 fma per tap, no LDS read
//   f_run1   4 x ds_read_b64 + 8 fma            per tap   (the shipped FMA16 loop's shape)
//   f_run2   5 x ds_read_b64 + 16 fma           per 2 taps
//   f_run3   6 x ds_read_b64 + 24 fma           per 3 taps
//   f_run4   7 x ds_read_b64 + 32 fma           per 4 taps
// Prints shader cycles per wave-TAP on a SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// one tap on rows r .. r+3 of buffer `b` (registers b + 2 r ...): 8 packed fmas, weight from s21 (low half)
#define XS(x) #x
#define FMA8_(b0, b1, b2, b3, b4, b5, b6, b7) \
  "v_pk_fma_f16 %0, s21, v" XS(b0) ", %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f16 %1, s21, v" XS(b1) ", %1 op_sel_hi:[0,1,1]\n\t" \
  "v_pk_fma_f16 %2, s21, v" XS(b2) ", %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f16 %3, s21, v" XS(b3) ", %3 op_sel_hi:[0,1,1]\n\t" \
  "v_pk_fma_f16 %4, s21, v" XS(b4) ", %4 op_sel_hi:[0,1,1]\n\tv_pk_fma_f16 %5, s21, v" XS(b5) ", %5 op_sel_hi:[0,1,1]\n\t" \
  "v_pk_fma_f16 %6, s21, v" XS(b6) ", %6 op_sel_hi:[0,1,1]\n\tv_pk_fma_f16 %7, s21, v" XS(b7) ", %7 op_sel_hi:[0,1,1]\n\t"
// buffer X = v32..v45 (7 rows), buffer Y = v46..v59; address v60
#define TAPX0 FMA8_(32, 33, 34, 35, 36, 37, 38, 39)
#define TAPX1 FMA8_(34, 35, 36, 37, 38, 39, 40, 41)
#define TAPX2 FMA8_(36, 37, 38, 39, 40, 41, 42, 43)
#define TAPX3 FMA8_(38, 39, 40, 41, 42, 43, 44, 45)
#define TAPY0 FMA8_(46, 47, 48, 49, 50, 51, 52, 53)
#define TAPY1 FMA8_(48, 49, 50, 51, 52, 53, 54, 55)
#define TAPY2 FMA8_(50, 51, 52, 53, 54, 55, 56, 57)
#define TAPY3 FMA8_(52, 53, 54, 55, 56, 57, 58, 59)
#define RDROW(b, k, off) "ds_read_b64 v[" XS(b) "+" XS(k) "*2:" XS(b) "+" XS(k) "*2+1], v60 offset:" XS(off) "\n\t"
#define ADDR "v_add_u32 v60, s20, %9\n\t"
#define RD4(b) ADDR RDROW(b, 0, 0) RDROW(b, 1, 448) RDROW(b, 2, 896) RDROW(b, 3, 1344)
#define RD5(b) RD4(b) RDROW(b, 4, 1792)
#define RD6(b) RD5(b) RDROW(b, 5, 2240)
#define RD7(b) RD6(b) RDROW(b, 6, 2688)
#define RD0(b) ""
#define STEP "s_waitcnt lgkmcnt(0)\n\ts_add_u32 s20, s20, 8\n\ts_and_b32 s20, s20, 0xbf\n\t"
#define KERNEL(NAME, RD, AX, AY)                                                                            \
  __global__ __launch_bounds__(256, 8) void NAME(unsigned *out, int groups) {                                        \
    extern __shared__ unsigned lds[];                                                                                \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                                      \
    for (int i = threadIdx.x; i < 19712 / 4; i += 256) lds[i] = 0x3c003c00u;                                         \
    __syncthreads();                                                                                                 \
    unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;                                         \
    const unsigned w = 0x1c001c00u;                                                                                  \
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds +                          \
                          (unsigned)((wave * 8 + (lane >> 5) * 4) * 448 + (lane & 31) * 8);                          \
    unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane(groups / 2);                                             \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();              \
    asm volatile("s_mov_b32 s20, 0\n\ts_mov_b32 s21, 0x1c001c00\n\t" RD(32) "L" #NAME "%=:\n\t"                        \
                 STEP RD(46) AX STEP RD(32) AY                                                                       \
                 "s_sub_u32 %10, %10, 1\n\ts_cmp_lg_u32 %10, 0\n\ts_cbranch_scc1 L" #NAME "%=\n\ts_waitcnt lgkmcnt(0)"       \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                     \
                 : "v"(w), "v"(base), "s"(cnt)                                                                         \
                 : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", \
                   "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "s20", "s21", "scc", "memory"); \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();              \
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 4] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; out[(blockIdx.x * 4 + wave) * 4 + 1] = (unsigned)(c1 - c0); out[(blockIdx.x * 4 + wave) * 4 + 2] = (unsigned)(r1 - r0); } \
  }
KERNEL(f_none, RD0, TAPX0, TAPY0)
KERNEL(f_run1, RD4, TAPX0, TAPY0)
KERNEL(f_run2, RD5, TAPX0 TAPX1, TAPY0 TAPY1)
KERNEL(f_run3, RD6, TAPX0 TAPX1 TAPX2, TAPY0 TAPY1 TAPY2)
KERNEL(f_run4, RD7, TAPX0 TAPX1 TAPX2 TAPX3, TAPY0 TAPY1 TAPY2 TAPY3)

int main() {
  unsigned *out; CHECK(hipMalloc(&out, 2048 * 4 * 4 * 4)); std::vector<unsigned> h(2048 * 16);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int taps = 480;
#define RUN(K, N) { const int groups = taps / N; for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(K, dim3(2048), dim3(256), 19712, 0, out, groups); \
    CHECK(hipEventRecord(e0, 0)); for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(K, dim3(2048), dim3(256), 19712, 0, out, groups); \
    CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize()); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20; \
    CHECK(hipMemcpy(h.data(), out, 2048 * 16 * 4, hipMemcpyDeviceToHost)); std::vector<double> clk; \
    for (int i = 0; i < 2048 * 4; ++i) clk.push_back(h[i * 4 + 1] / (h[i * 4 + 2] * 10.0)); \
    std::sort(clk.begin(), clk.end()); const double ghz = clk[clk.size() / 2]; \
    printf("%-8s %.2f us per launch of %d taps x 8 waves per SIMD = %.1f ns per wave-tap = %.1f cycles at the measured %.2f GHz\n", #K, ms * 1e3, taps, ms * 1e6 / (taps * 8.0), ms * 1e6 / (taps * 8.0) * ghz, ghz); }
  RUN(f_none, 1) RUN(f_run1, 1) RUN(f_run2, 2) RUN(f_run3, 3) RUN(f_run4, 4) RUN(f_none, 1) RUN(f_run1, 1) RUN(f_run2, 2) RUN(f_run3, 3) RUN(f_run4, 4)
  return 0;
}
