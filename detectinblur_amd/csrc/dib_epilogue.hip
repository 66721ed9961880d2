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


// ---- the same epilogue for batches whose images need the detector's internal resize (every real COCO batch: the reference
// blurs at native size, engine.py:101, and resizes inside the model, net_transforms.py:151-175) ------------------------------
// out[y, x] = bilinear(normalised image) exactly as torch.nn.functional.interpolate(..., mode="bilinear",
// recompute_scale_factor=True, align_corners=False) computes it on the GPU (ATen UpSampleBilinear2d.cu):
//   rheight = float(H) / float(Ho)                      (the scale recomputed from the integer sizes, host side)
//   h1r = max(rheight * (y + 0.5) - 0.5, 0);  h1 = int(h1r);  h1p = h1 < H - 1;  l1 = h1r - h1;  l0 = 1 - l1     (same for x)
//   val = l0h * (l0w * p[h1][w1] + l1w * p[h1][w1 + w1p]) + l1h * (l0w * p[h1 + h1p][w1] + l1w * p[h1 + h1p][w1 + w1p])
// ATen's kernels are built with hipcc's default floating-point contraction, so `a * b + c * d` there is fma(a, b, c * d) and
// `s * t - 0.5` is fma(s, t, -0.5): FMA = true restates exactly that (bit-identical to the eager GPU path,
// tests/test_epilogue_gpu.py); FMA = false is the uncontracted expression (the CPU's arithmetic).  The four source pixels
// are normalised first, ((float)p - mean) / std, which is what the reference interpolates.  Images whose output size equals
// their input size take no interpolation at all (the reference skips the call at scale factor 1).
struct ResizeBatch {
  const void *in[MAX_BATCH];
  int H[MAX_BATCH], W[MAX_BATCH], Ho[MAX_BATCH], Wo[MAX_BATCH];
  float rh[MAX_BATCH], rw[MAX_BATCH];
  float mean[MAX_BATCH][4], std[MAX_BATCH][4];
};

template <bool FMA> __device__ __forceinline__ float src_index(float scale, int dst) {
#pragma clang fp contract(off)
  const float t = (float)dst + 0.5f;
  const float s = FMA ? __builtin_fmaf(scale, t, -0.5f) : scale * t - 0.5f;
  return s < 0.f ? 0.f : s;
}

template <bool FMA> __device__ __forceinline__ float lerp2(float l0h, float l1h, float l0w, float l1w, float p00, float p01, float p10, float p11) {
#pragma clang fp contract(off)
  if (FMA) {
    const float top = __builtin_fmaf(l0w, p00, l1w * p01), bot = __builtin_fmaf(l0w, p10, l1w * p11);
    return __builtin_fmaf(l0h, top, l1h * bot);
  }
  return l0h * (l0w * p00 + l1w * p01) + l1h * (l0w * p10 + l1w * p11);
}

template <typename T, bool NHWC, bool FMA>
__global__ __launch_bounds__(256) void normalize_resize_pad_kernel(ResizeBatch b, float *__restrict__ out, int Hp, int Wp) {
#pragma clang fp contract(off)
  const int img = blockIdx.z;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= Wp) return;
  const int H = b.H[img], W = b.W[img], Ho = b.Ho[img], Wo = b.Wo[img];
  float v[3] = {0.f, 0.f, 0.f};
  if (y < Ho && x < Wo) {
    const T *p = reinterpret_cast<const T *>(b.in[img]);
    const size_t plane = (size_t)H * W;
    if (Ho == H && Wo == W) {
      const T *q = p + (size_t)y * W + x;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = ((float)q[c * plane] - b.mean[img][c]) / b.std[img][c];
    } else {
      const float h1r = src_index<FMA>(b.rh[img], y), w1r = src_index<FMA>(b.rw[img], x);
      const int h1 = (int)h1r, w1 = (int)w1r;
      const int h1p = h1 < H - 1 ? 1 : 0, w1p = w1 < W - 1 ? 1 : 0;
      const float l1h = h1r - (float)h1, l0h = 1.f - l1h, l1w = w1r - (float)w1, l0w = 1.f - l1w;
      const T *q0 = p + (size_t)h1 * W + w1, *q1 = q0 + (size_t)h1p * W;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float m = b.mean[img][c], sd = b.std[img][c];
        const float p00 = ((float)q0[c * plane] - m) / sd, p01 = ((float)q0[c * plane + w1p] - m) / sd;
        const float p10 = ((float)q1[c * plane] - m) / sd, p11 = ((float)q1[c * plane + w1p] - m) / sd;
        v[c] = lerp2<FMA>(l0h, l1h, l0w, l1w, p00, p01, p10, p11);
      }
    }
  }
  if (NHWC) {
    float *o = out + (((size_t)img * Hp + y) * Wp + x) * 3;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
  } else {
    const size_t plane = (size_t)Hp * Wp;
    float *o = out + (size_t)img * 3 * plane + (size_t)y * Wp + x;
    o[0] = v[0]; o[plane] = v[1]; o[2 * plane] = v[2];
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

// 1 (default): ATen's contraction pattern (bit-identical to the eager GPU path); 0: uncontracted.  Test hook.
static int g_resize_fma = 1;
extern "C" void dib_debug_set_resize_contraction(int on) { g_resize_fma = on ? 1 : 0; }

extern "C" int dib_normalize_resize_pad(const void *const *in_dev, int dtype, const int *H, const int *W, const int *Ho, const int *Wo, int B,
                                        const float *mean, const float *std, float *out_dev, int Hp, int Wp, int channels_last,
                                        void *stream) {
  if (B < 0 || (B > 0 && (!in_dev || !H || !W || !Ho || !Wo || !mean || !std || !out_dev))) { set_error("dib_normalize_resize_pad: null pointer or negative batch"); return DIB_EINVAL; }
  if (dtype != DIB_F16 && dtype != DIB_F32) { set_error("dib_normalize_resize_pad: unknown dtype %d", dtype); return DIB_EINVAL; }
  if (Hp <= 0 || Wp <= 0) { set_error("dib_normalize_resize_pad: empty batch shape"); return DIB_EINVAL; }
  for (int i = 0; i < B; ++i) {
    if (!in_dev[i] || H[i] <= 0 || W[i] <= 0 || Ho[i] <= 0 || Wo[i] <= 0 || Ho[i] > Hp || Wo[i] > Wp) {
      set_error("dib_normalize_resize_pad: image %d is null, empty or (resized) larger than the batch", i);
      return DIB_EINVAL;
    }
    if ((uintptr_t)in_dev[i] & (dtype == DIB_F16 ? 1 : 3)) { set_error("dib_normalize_resize_pad: image %d is misaligned", i); return DIB_EINVAL; }
  }
  hipStream_t s = (hipStream_t)stream;
  for (int b0 = 0; b0 < B; b0 += MAX_BATCH) {
    const int n = B - b0 < MAX_BATCH ? B - b0 : MAX_BATCH;
    ResizeBatch rb;
    for (int i = 0; i < n; ++i) {
      const int k = b0 + i;
      rb.in[i] = in_dev[k]; rb.H[i] = H[k]; rb.W[i] = W[k]; rb.Ho[i] = Ho[k]; rb.Wo[i] = Wo[k];
      rb.rh[i] = (float)H[k] / (float)Ho[k];      // ATen: area_pixel_compute_scale<float>(input, output, align_corners = false, no scale)
      rb.rw[i] = (float)W[k] / (float)Wo[k];
      for (int c = 0; c < 3; ++c) { rb.mean[i][c] = mean[k * 3 + c]; rb.std[i][c] = std[k * 3 + c]; }
      rb.mean[i][3] = 0.f; rb.std[i][3] = 1.f;
    }
    float *out = out_dev + (size_t)b0 * 3 * Hp * Wp;
    const dim3 grid((Wp + 255) / 256, Hp, n), block(256);
#define DIB_LAUNCH_RESIZE(T, CL, FM) hipLaunchKernelGGL((normalize_resize_pad_kernel<T, CL, FM>), grid, block, 0, s, rb, out, Hp, Wp)
    if (dtype == DIB_F16) {
      if (channels_last) { if (g_resize_fma) DIB_LAUNCH_RESIZE(_Float16, true, true); else DIB_LAUNCH_RESIZE(_Float16, true, false); }
      else { if (g_resize_fma) DIB_LAUNCH_RESIZE(_Float16, false, true); else DIB_LAUNCH_RESIZE(_Float16, false, false); }
    } else {
      if (channels_last) { if (g_resize_fma) DIB_LAUNCH_RESIZE(float, true, true); else DIB_LAUNCH_RESIZE(float, true, false); }
      else { if (g_resize_fma) DIB_LAUNCH_RESIZE(float, false, true); else DIB_LAUNCH_RESIZE(float, false, false); }
    }
#undef DIB_LAUNCH_RESIZE
  }
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
