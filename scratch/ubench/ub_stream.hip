// Micro-benchmark (not product code): in-place bias + ReLU epilogue over a 550 MB channels-last tensor (8 x 256 x 200 x 336 fp32),
// the shape of dib_eltwise.hip's bias_act_vec4_kernel: grid-stride with a capped grid vs one float4 per thread, 64-bit vs 32-bit
// index arithmetic, with and without the sign mask.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool MASK>
__global__ __launch_bounds__(256) void k_stride64(float4 *x, const float4 *bias, long long n4, int C4, unsigned char *mask) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = x[i]; const float4 b = bias[(int)(i % C4)];
    v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
    x[i] = v;
    if (MASK) mask[i] = (unsigned char)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
  }
}
template <bool MASK>
__global__ __launch_bounds__(256) void k_stride32(float4 *x, const float4 *bias, unsigned n4, unsigned C4, unsigned char *mask) {
  const unsigned step = gridDim.x * blockDim.x;
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned c = i % C4;
  const unsigned cstep = step % C4;
  for (; i < n4; i += step) {
    float4 v = x[i]; const float4 b = bias[c];
    c += cstep; if (c >= C4) c -= C4;
    v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
    x[i] = v;
    if (MASK) mask[i] = (unsigned char)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
  }
}
template <bool MASK>
__global__ __launch_bounds__(256) void k_one32(float4 *x, const float4 *bias, unsigned n4, unsigned C4, unsigned char *mask) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = x[i]; const float4 b = bias[i % C4];
  v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
  x[i] = v;
  if (MASK) mask[i] = (unsigned char)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
}
// two float4 per thread, both loads issued before the arithmetic
template <bool MASK>
__global__ __launch_bounds__(256) void k_two32(float4 *x, const float4 *bias, unsigned n4, unsigned C4, unsigned char *mask) {
  const unsigned i = (blockIdx.x * blockDim.x * 2) + threadIdx.x;
  const unsigned j = i + blockDim.x;
  if (j >= n4) { if (i < n4) { float4 v = x[i]; const float4 b = bias[i % C4]; v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f); x[i] = v; if (MASK) mask[i] = 0; } return; }
  float4 v = x[i], u = x[j]; const float4 b = bias[i % C4], d = bias[j % C4];
  v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
  u.x = fmaxf(u.x + d.x, 0.f); u.y = fmaxf(u.y + d.y, 0.f); u.z = fmaxf(u.z + d.z, 0.f); u.w = fmaxf(u.w + d.w, 0.f);
  x[i] = v; x[j] = u;
  if (MASK) { mask[i] = (unsigned char)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
              mask[j] = (unsigned char)((u.x > 0.f ? 1 : 0) | (u.y > 0.f ? 2 : 0) | (u.z > 0.f ? 4 : 0) | (u.w > 0.f ? 8 : 0)); }
}

// ATen's elementwise shape: 128 threads, 4 scalar elements per thread, strided by the block (every load a coalesced 512 B per wave)
template <int VEC>   // VEC floats per access (1, 2 or 4), 4 accesses per thread
__global__ __launch_bounds__(128) void k_unroll4(float *x, const float *bias, unsigned n, unsigned C) {
  const unsigned base = blockIdx.x * (128u * 4u * VEC) + threadIdx.x * VEC;
  float v[4][VEC];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const unsigned i = base + u * 128u * VEC;
    if (i < n) {
      if (VEC == 4) { const float4 t = *reinterpret_cast<const float4 *>(x + i); v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
      else if (VEC == 2) { const float2 t = *reinterpret_cast<const float2 *>(x + i); v[u][0] = t.x; v[u][1] = t.y; }
      else v[u][0] = x[i];
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const unsigned i = base + u * 128u * VEC;
    if (i < n) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[u][k] = fmaxf(v[u][k] + bias[(i + k) % C], 0.f);
      if (VEC == 4) *reinterpret_cast<float4 *>(x + i) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
      else if (VEC == 2) *reinterpret_cast<float2 *>(x + i) = make_float2(v[u][0], v[u][1]);
      else x[i] = v[u][0];
    }
  }
}

int main() {
  const long long n = 8ll * 256 * 200 * 336, n4 = n / 4; const int C4 = 64;
  float4 *x, *bias; unsigned char *mask;
  CHECK(hipMalloc(&x, n * 4)); CHECK(hipMalloc(&bias, 1024)); CHECK(hipMalloc(&mask, n4));
  CHECK(hipMemset(x, 0, n * 4)); CHECK(hipMemset(bias, 0, 1024));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double bytes = (double)n * 8;
#define TIME(NAME, LAUNCH) { auto fn = LAUNCH; for (int i = 0; i < 3; ++i) fn(); CHECK(hipEventRecord(e0, 0)); for (int i = 0; i < 20; ++i) fn(); CHECK(hipEventRecord(e1, 0)); \
    CHECK(hipDeviceSynchronize()); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); printf("%-34s %7.1f us  %.2f TB/s (x read + written)\n", NAME, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12); }
  const unsigned full = (unsigned)((n4 + 255) / 256);
  for (int rep = 0; rep < 2; ++rep) {
    TIME("stride64 cap 8192, no mask", [&] { hipLaunchKernelGGL(k_stride64<false>, dim3(8192), dim3(256), 0, 0, x, bias, n4, C4, mask); });
    TIME("stride64 cap 8192, mask", [&] { hipLaunchKernelGGL(k_stride64<true>, dim3(8192), dim3(256), 0, 0, x, bias, n4, C4, mask); });
    TIME("stride32 cap 8192, no mask", [&] { hipLaunchKernelGGL(k_stride32<false>, dim3(8192), dim3(256), 0, 0, x, bias, (unsigned)n4, C4, mask); });
    TIME("stride32 cap 8192, mask", [&] { hipLaunchKernelGGL(k_stride32<true>, dim3(8192), dim3(256), 0, 0, x, bias, (unsigned)n4, C4, mask); });
    TIME("stride32 cap 2048, mask", [&] { hipLaunchKernelGGL(k_stride32<true>, dim3(2048), dim3(256), 0, 0, x, bias, (unsigned)n4, C4, mask); });
    TIME("stride32 cap 32768, mask", [&] { hipLaunchKernelGGL(k_stride32<true>, dim3(32768), dim3(256), 0, 0, x, bias, (unsigned)n4, C4, mask); });
    TIME("one float4 per thread, no mask", [&] { hipLaunchKernelGGL(k_one32<false>, dim3(full), dim3(256), 0, 0, x, bias, (unsigned)n4, C4, mask); });
    TIME("one float4 per thread, mask", [&] { hipLaunchKernelGGL(k_one32<true>, dim3(full), dim3(256), 0, 0, x, bias, (unsigned)n4, C4, mask); });
    TIME("128 thr x 4 x float (ATen shape)", [&] { hipLaunchKernelGGL(k_unroll4<1>, dim3((unsigned)((n + 511) / 512)), dim3(128), 0, 0, (float *)x, (const float *)bias, (unsigned)n, 256u); });
    TIME("128 thr x 4 x float2", [&] { hipLaunchKernelGGL(k_unroll4<2>, dim3((unsigned)((n + 1023) / 1024)), dim3(128), 0, 0, (float *)x, (const float *)bias, (unsigned)n, 256u); });
    TIME("128 thr x 4 x float4", [&] { hipLaunchKernelGGL(k_unroll4<4>, dim3((unsigned)((n + 2047) / 2048)), dim3(128), 0, 0, (float *)x, (const float *)bias, (unsigned)n, 256u); });
    TIME("two float4 per thread, no mask", [&] { hipLaunchKernelGGL(k_two32<false>, dim3((full + 1) / 2), dim3(256), 0, 0, x, bias, (unsigned)n4, C4, mask); });
    TIME("two float4 per thread, mask", [&] { hipLaunchKernelGGL(k_two32<true>, dim3((full + 1) / 2), dim3(256), 0, 0, x, bias, (unsigned)n4, C4, mask); });
  }
  return 0;
}
