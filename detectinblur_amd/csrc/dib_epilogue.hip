// Fused epilogue of the blur for the detector's input: the blurred fp16 planes of a
// batch -> fp32, per-image normalisation, zero-padded batch tensor, in ONE pass.  It replaces, when no resize is
// needed (BASELINE's 800 x 1333 images: scale factor exactly 1), the reference's
//   image.float()                              engine.py:107-110         (read 2 B, write 4 B per element)
//   (image - mean[:, None, None]) / std[...]   net_transforms.py:135-139 (read 4 + write 4, twice in eager PyTorch)
//   batched_imgs.new_full(0); pad_img.copy_()  net_transforms.py:238-247 (write 4, read 4, write 4)
// whose six passes move 34 B per element; this kernel moves 2 + 4 = 6 B per element plus the padding zeros:
// algorithmic bytes per 3 x 800 x 1333 image = 6,398,400 read + 12,796,800 written = 19,195,200 B (HBM-bound).
// Arithmetic = the reference's: float(x), subtract the fp32 mean, IEEE fp32 divide by the fp32 std (no reciprocal
// multiply, no contraction): bit-identical to the unfused path (tests/test_epilogue_gpu.py).
#include "dib_common.h"
#include <hip/hip_fp16.h>

namespace dib {

struct EpilogueBatch {
  const void *in[MAX_BATCH];
  int H[MAX_BATCH], W[MAX_BATCH];
  float mean[MAX_BATCH][4], std[MAX_BATCH][4];   // per image, channels 0..2 (a 4th slot pads to 16 B)
};

// One thread per output pixel (all C = 3 channels): the three planes are read with lane-consecutive 2- or 4-byte
// loads, the channels-last output is one 12-byte store per lane (768 contiguous bytes per wave); the planar output
// three lane-consecutive 4-byte stores.  Pixels outside the image (bottom / right padding) are written as zeros.
template <typename T, bool NHWC>
__global__ __launch_bounds__(256) void normalize_pad_kernel(EpilogueBatch b, float *__restrict__ out, int Hp, int Wp) {
#pragma clang fp contract(off)
  const int img = blockIdx.z;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= Wp) return;
  const int H = b.H[img], W = b.W[img];
  float v0 = 0.f, v1 = 0.f, v2 = 0.f;
  if (y < H && x < W) {
    const T *p = reinterpret_cast<const T *>(b.in[img]) + (size_t)y * W + x;
    const size_t plane = (size_t)H * W;
    v0 = ((float)p[0] - b.mean[img][0]) / b.std[img][0];
    v1 = ((float)p[plane] - b.mean[img][1]) / b.std[img][1];
    v2 = ((float)p[2 * plane] - b.mean[img][2]) / b.std[img][2];
  }
  if (NHWC) {
    float *o = out + (((size_t)img * Hp + y) * Wp + x) * 3;
    o[0] = v0; o[1] = v1; o[2] = v2;
  } else {
    const size_t plane = (size_t)Hp * Wp;
    float *o = out + (size_t)img * 3 * plane + (size_t)y * Wp + x;
    o[0] = v0; o[plane] = v1; o[2 * plane] = v2;
  }
}

}  // namespace dib

using namespace dib;

extern "C" int dib_normalize_pad(const void *const *in_dev, int dtype, const int *H, const int *W, int B, const float *mean,
                                 const float *std, float *out_dev, int Hp, int Wp, int channels_last, void *stream) {
  if (B < 0 || (B > 0 && (!in_dev || !H || !W || !mean || !std || !out_dev))) { set_error("dib_normalize_pad: null pointer or negative batch"); return DIB_EINVAL; }
  if (dtype != DIB_F16 && dtype != DIB_F32) { set_error("dib_normalize_pad: unknown dtype %d", dtype); return DIB_EINVAL; }
  if (Hp <= 0 || Wp <= 0) { set_error("dib_normalize_pad: empty batch shape"); return DIB_EINVAL; }
  for (int i = 0; i < B; ++i) {
    if (!in_dev[i] || H[i] <= 0 || W[i] <= 0 || H[i] > Hp || W[i] > Wp) { set_error("dib_normalize_pad: image %d is null, empty or larger than the batch", i); return DIB_EINVAL; }
    if ((uintptr_t)in_dev[i] & (dtype == DIB_F16 ? 1 : 3)) { set_error("dib_normalize_pad: image %d is misaligned", i); return DIB_EINVAL; }
  }
  hipStream_t s = (hipStream_t)stream;
  for (int b0 = 0; b0 < B; b0 += MAX_BATCH) {
    const int n = B - b0 < MAX_BATCH ? B - b0 : MAX_BATCH;
    EpilogueBatch eb;
    for (int i = 0; i < n; ++i) {
      eb.in[i] = in_dev[b0 + i]; eb.H[i] = H[b0 + i]; eb.W[i] = W[b0 + i];
      for (int c = 0; c < 3; ++c) { eb.mean[i][c] = mean[(b0 + i) * 3 + c]; eb.std[i][c] = std[(b0 + i) * 3 + c]; }
      eb.mean[i][3] = 0.f; eb.std[i][3] = 1.f;
    }
    float *out = out_dev + (size_t)b0 * 3 * Hp * Wp;
    const dim3 grid((Wp + 255) / 256, Hp, n), block(256);
    if (dtype == DIB_F16) {
      if (channels_last) hipLaunchKernelGGL((normalize_pad_kernel<_Float16, true>), grid, block, 0, s, eb, out, Hp, Wp);
      else hipLaunchKernelGGL((normalize_pad_kernel<_Float16, false>), grid, block, 0, s, eb, out, Hp, Wp);
    } else {
      if (channels_last) hipLaunchKernelGGL((normalize_pad_kernel<float, true>), grid, block, 0, s, eb, out, Hp, Wp);
      else hipLaunchKernelGGL((normalize_pad_kernel<float, false>), grid, block, 0, s, eb, out, Hp, Wp);
    }
  }
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
