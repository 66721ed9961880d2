// Post-blur corruption chain of `manual_blur` (reference models/blur_functions.py:72-81) in ONE pass over the blurred image:
//   add_noise:  output = clamp(output + randn_like(output) * sqrt(noise_var), 0, 1)                       (:72-74)
//   add_block:  output = interpolate(interpolate(output, scale_factor=s, 'nearest'), size=original, 'nearest')   (:76-81)
// Eager PyTorch runs it as 5 + 2 launches with four full-size intermediates; here every output pixel gathers its source
// pixel through the composed nearest-neighbour maps (down, then up) and adds that SOURCE pixel's noise, which is a pure
// function of (seed, element index) -- a counter-based generator (Philox-4x32-10, Box-Muller) -- so duplicated blocks carry
// the same noise as they do when the noisy image is down/up-sampled.  HBM-bound: 2 (or 4) bytes read + written per element.
//
// Parity: the block path is index arithmetic and equals torch's two `interpolate` calls bit for bit
//   (ATen upsample_nearest2d: src = min(int(floorf(dst * scale)), in - 1), scale = float(1 / scale_factor) for the first
//   call, float(in) / out for the second; output size of the first = floor(double(in) * scale_factor));
//   the noise is drawn from this kernel's own generator: same distribution and the same rounding steps as torch's Half
//   expression (normal rounded to the image type, product and sum each rounded once), NOT torch's sample values
//   (torch's device generator does not reproduce its host generator either: tests check the statistics).
#include "dib_common.h"
#include <hip/hip_fp16.h>
#include <math.h>

namespace dib {

struct PostOps {
  const void *in;
  void *out;
  int C, H, W;
  int block;                 // 0: no block artefacts
  int hs, ws;                // size of the down-sampled image
  float down_h, down_w;      // float(1 / scale_factor): first interpolate
  float up_h, up_w;          // float(hs) / H, float(ws) / W: second interpolate
  int noise;                 // 0: no noise
  float noise_std;
  unsigned key0, key1;       // Philox key = seed
};

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
  c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
}

// One standard normal per (seed, index): Philox-4x32-10 on counter (index, 0, 0, 0), Box-Muller on its first two words.
__device__ __forceinline__ float normal_at(unsigned long long index, unsigned k0, unsigned k1) {
  unsigned c[4] = {(unsigned)index, (unsigned)(index >> 32), 0u, 0u};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);     // (0, 1)
  const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
}

template <typename T> struct Px;
template <> struct Px<_Float16> {
  static __device__ float add_noise(_Float16 v, float n, float s) {
    const _Float16 n16 = (_Float16)n;                      // randn_like(half)
    const _Float16 prod = (_Float16)((float)n16 * s);      // half * python scalar: fp32 opmath, one rounding
    return (float)(_Float16)((float)v + (float)prod);      // half + half
  }
};
template <> struct Px<float> {
  static __device__ float add_noise(float v, float n, float s) { return v + n * s; }
};

template <typename T>
__global__ __launch_bounds__(256) void post_ops_kernel(PostOps p) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= p.W) return;
  int sy = y, sx = x;
  if (p.block) {
    const int my = min((int)floorf((float)y * p.up_h), p.hs - 1), mx = min((int)floorf((float)x * p.up_w), p.ws - 1);
    sy = min((int)floorf((float)my * p.down_h), p.H - 1);
    sx = min((int)floorf((float)mx * p.down_w), p.W - 1);
  }
  const size_t plane = (size_t)p.H * p.W;
  const T *in = reinterpret_cast<const T *>(p.in);
  T *out = reinterpret_cast<T *>(p.out);
  for (int c = 0; c < p.C; ++c) {
    const size_t src = c * plane + (size_t)sy * p.W + sx;
    float v = (float)in[src];
    if (p.noise) {
      v = Px<T>::add_noise(in[src], normal_at(src, p.key0, p.key1), p.noise_std);
      v = fminf(fmaxf(v, 0.f), 1.f);
    }
    out[c * plane + (size_t)y * p.W + x] = (T)v;
  }
}

}  // namespace dib

using namespace dib;

// in_dev / out_dev: C x H x W planes of `dtype` (DIB_F16 / DIB_F32), out must not alias in (the block path gathers).
// noise_var <= 0: no noise; block_scale <= 0: no block artefacts.  seed: the noise field's key.
extern "C" int dib_post_ops(const void *in_dev, void *out_dev, int C, int H, int W, int dtype, double noise_var,
                            unsigned long long seed, double block_scale, void *stream) {
  if (!in_dev || !out_dev) { set_error("dib_post_ops: null pointer"); return DIB_EINVAL; }
  if (in_dev == out_dev) { set_error("dib_post_ops: out aliases in"); return DIB_EINVAL; }
  if (C <= 0 || H <= 0 || W <= 0) { set_error("dib_post_ops: empty image"); return DIB_EINVAL; }
  if (dtype != DIB_F16 && dtype != DIB_F32) { set_error("dib_post_ops: unknown dtype %d", dtype); return DIB_EINVAL; }
  PostOps p;
  p.in = in_dev; p.out = out_dev; p.C = C; p.H = H; p.W = W;
  p.noise = noise_var > 0.0;
  p.noise_std = (float)sqrt(noise_var > 0.0 ? noise_var : 0.0);
  p.key0 = (unsigned)seed; p.key1 = (unsigned)(seed >> 32);
  p.block = block_scale > 0.0;
  p.hs = H; p.ws = W; p.down_h = p.down_w = p.up_h = p.up_w = 1.f;
  if (p.block) {
    p.hs = (int)floor((double)H * block_scale);           // torch.nn.functional.interpolate: floor(float(size) * scale_factor)
    p.ws = (int)floor((double)W * block_scale);
    if (p.hs <= 0 || p.ws <= 0) { set_error("dib_post_ops: scale factor %g leaves no pixels", block_scale); return DIB_ESHAPE; }
    p.down_h = p.down_w = (float)(1.0 / block_scale);
    p.up_h = (float)p.hs / (float)H;
    p.up_w = (float)p.ws / (float)W;
  }
  const dim3 grid((W + 255) / 256, H);
  if (dtype == DIB_F16) hipLaunchKernelGGL(post_ops_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(post_ops_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
