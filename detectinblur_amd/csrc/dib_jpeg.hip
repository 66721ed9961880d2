// JPEG round trip of the post-blur corruption chain (`--add_jpeg_artefacts`) in ONE launch: the reference's
// transforms.add_jpeg_artifact_to_image (transforms.py:467-493: reflect-pad to a multiple of 16, DiffJPEG, crop) around
// models/jpeg/DiffJPEG (compression.py / decompression.py): x 255 -> YCbCr -> 4:2:0 chroma (2x2 mean) -> 8x8 blocks ->
// DCT-II of (block - 128) -> divide by (table * factor), round half to even -> multiply back -> inverse DCT + 128 ->
// chroma repeated 2x2 -> RGB -> clamp [0, 255] -> / 255.  Stock torch runs it as ~30 launches with a dozen full-size
// intermediates; here one workgroup owns one 16 x 16 macroblock (4 luma + 2 chroma blocks) from the padded load to the
// cropped store, everything in between in LDS.
//
// Parity: same operations in fp32; the 64-term DCT sums run in this kernel's own order, so a coefficient within rounding
// noise of .5 can land on the other side and move one quantisation step -- the tolerance the torch path on the GPU already
// has against the reference's CPU output (tests/test_jpeg.py: mean abs 2e-3, max one luminance step).
#include "dib_common.h"
#include <hip/hip_fp16.h>
#include <math.h>

namespace dib {

struct JpegArgs {
  const void *in;     // 3 x H x W planes
  void *out;          // 3 x H x W planes of fp16 (the reference returns `.half()`)
  int H, W;
  int top, left;      // reflect padding in front (transforms.py:471-473)
  int Hp, Wp;         // padded size, multiples of 16
  float qy[64], qc[64];   // table * factor, indexed [u][v] like the module's buffers
};

__device__ __forceinline__ int reflect(int s, int n) {      // F.pad(mode='reflect'): no edge repeat
  s = s < 0 ? -s : s;
  return s > n - 1 ? 2 * (n - 1) - s : s;
}

template <typename T>
__global__ __launch_bounds__(256) void jpeg_roundtrip_kernel(JpegArgs p) {
#pragma clang fp contract(off)
  __shared__ float ycc[3][16][16];      // Y, Cb, Cr of the macroblock
  __shared__ float blk[6][8][8];        // the six 8x8 blocks: pixels, then coefficients, then pixels again
  __shared__ float coef[6][8][8];
  __shared__ float ct[8][8];            // ct[a][u] = cos((2a+1) u pi / 16)
  const int t = threadIdx.x, ly = t >> 4, lx = t & 15;
  const int py = blockIdx.y * 16 + ly, px = blockIdx.x * 16 + lx;        // padded coordinates
  if (t < 64) ct[t >> 3][t & 7] = cosf((float)((2 * (t >> 3) + 1) * (t & 7)) * 0.19634954084936207f);
  // ---- load + colour transform (compression.py rgb_to_ycbcr_jpeg: image * 255 . matrix + shift) --------------------
  const int sy = reflect(py - p.top, p.H), sx = reflect(px - p.left, p.W);
  const size_t plane = (size_t)p.H * p.W, src = (size_t)sy * p.W + sx;
  const T *in = reinterpret_cast<const T *>(p.in);
  const float r = (float)in[src] * 255.f, g = (float)in[plane + src] * 255.f, b = (float)in[2 * plane + src] * 255.f;
  ycc[0][ly][lx] = r * 0.299f + g * 0.587f + b * 0.114f;
  ycc[1][ly][lx] = r * -0.168736f + g * -0.331264f + b * 0.5f + 128.f;
  ycc[2][ly][lx] = r * 0.5f + g * -0.418688f + b * -0.081312f + 128.f;
  __syncthreads();
  // ---- blocks: luma 0..3 = (row half, column half); chroma 4, 5 = 2x2 means (avg_pool2d) -----------------------------
  {
    const int bi = (ly >> 3) * 2 + (lx >> 3);
    blk[bi][ly & 7][lx & 7] = ycc[0][ly][lx] - 128.f;
    if (t < 128) {
      const int c = t >> 6, yy = (t >> 3) & 7, xx = t & 7;
      const float s = ycc[1 + c][2 * yy][2 * xx] + ycc[1 + c][2 * yy][2 * xx + 1] + ycc[1 + c][2 * yy + 1][2 * xx] + ycc[1 + c][2 * yy + 1][2 * xx + 1];
      blk[4 + c][yy][xx] = s * 0.25f - 128.f;
    }
  }
  __syncthreads();
  // ---- forward DCT, quantise, dequantise: coefficient (u, v) of block k by thread k*64 + u*8 + v (two rounds) ---------
  for (int k0 = 0; k0 < 6; k0 += 4) {
    const int k = k0 + (t >> 6), u = (t >> 3) & 7, v = t & 7;
    if (k < 6) {
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        float row = 0.f;
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) row += blk[k][a][bb] * ct[bb][v];
        acc += row * ct[a][u];
      }
      const float alpha = (u == 0 ? 0.70710678118654752f : 1.f) * (v == 0 ? 0.70710678118654752f : 1.f);
      const float c = acc * (alpha * 0.25f);
      const float q = k < 4 ? p.qy[u * 8 + v] : p.qc[u * 8 + v];
      coef[k][u][v] = rintf(c / q) * q * alpha;          // torch.round = round half to even; idct multiplies by alpha again
    }
  }
  __syncthreads();
  // ---- inverse DCT + 128 (decompression.py idct_8x8: 0.25 * sum alpha coef cos cos + 128) -----------------------------
  for (int k0 = 0; k0 < 6; k0 += 4) {
    const int k = k0 + (t >> 6), a = (t >> 3) & 7, bb = t & 7;
    if (k < 6) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float row = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v) row += coef[k][u][v] * ct[bb][v];
        acc += row * ct[a][u];
      }
      blk[k][a][bb] = acc * 0.25f + 128.f;
    }
  }
  __syncthreads();
  // ---- chroma repeated 2x2, YCbCr -> RGB, clamp, / 255, cropped store ---------------------------------------------------
  const int oy = py - p.top, ox = px - p.left;
  if (oy < 0 || oy >= p.H || ox < 0 || ox >= p.W) return;
  const float Y = blk[(ly >> 3) * 2 + (lx >> 3)][ly & 7][lx & 7];
  const float Cb = blk[4][ly >> 1][lx >> 1] - 128.f, Cr = blk[5][ly >> 1][lx >> 1] - 128.f;
  float R = Y + Cr * 1.402f;
  float G = Y + Cb * -0.344136f + Cr * -0.714136f;
  float B = Y + Cb * 1.772f;
  R = fminf(255.f, fmaxf(0.f, R)) / 255.f;
  G = fminf(255.f, fmaxf(0.f, G)) / 255.f;
  B = fminf(255.f, fmaxf(0.f, B)) / 255.f;
  _Float16 *out = reinterpret_cast<_Float16 *>(p.out);
  const size_t dst = (size_t)oy * p.W + ox;
  out[dst] = (_Float16)R; out[plane + dst] = (_Float16)G; out[2 * plane + dst] = (_Float16)B;
}

}  // namespace dib

using namespace dib;

// in_dev: 3 x H x W planes of `dtype` in [0, 1]; out_dev: 3 x H x W planes of fp16; q_luma / q_chroma: the 8 x 8 tables
// ALREADY multiplied by the quality factor, row-major [u][v] as models/jpeg.py holds them (host pointers, 64 floats each).
extern "C" int dib_jpeg_roundtrip(const void *in_dev, void *out_dev, int H, int W, int dtype, const float *q_luma,
                                  const float *q_chroma, void *stream) {
  if (!in_dev || !out_dev || !q_luma || !q_chroma) { set_error("dib_jpeg_roundtrip: null pointer"); return DIB_EINVAL; }
  if (H < 2 || W < 2) { set_error("dib_jpeg_roundtrip: image too small for reflect padding"); return DIB_ESHAPE; }
  if (dtype != DIB_F16 && dtype != DIB_F32) { set_error("dib_jpeg_roundtrip: unknown dtype %d", dtype); return DIB_EINVAL; }
  JpegArgs p;
  p.in = in_dev; p.out = out_dev; p.H = H; p.W = W;
  const int wp = 16 - W % 16, hp = 16 - H % 16;              // transforms.py:471-472 (a full 16 when already a multiple)
  p.left = wp / 2; p.top = hp / 2;
  p.Wp = W + wp; p.Hp = H + hp;
  if (p.top > H - 1 || hp - p.top > H - 1 || p.left > W - 1 || wp - p.left > W - 1) {
    set_error("dib_jpeg_roundtrip: padding size should be less than the corresponding input dimension (%d x %d)", H, W);
    return DIB_ESHAPE;
  }
  for (int i = 0; i < 64; ++i) {
    if (!(q_luma[i] > 0.f) || !(q_chroma[i] > 0.f)) { set_error("dib_jpeg_roundtrip: quantisation tables must be positive"); return DIB_EINVAL; }
    p.qy[i] = q_luma[i]; p.qc[i] = q_chroma[i];
  }
  const dim3 grid(p.Wp / 16, p.Hp / 16);
  if (dtype == DIB_F16) hipLaunchKernelGGL(jpeg_roundtrip_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(jpeg_roundtrip_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
