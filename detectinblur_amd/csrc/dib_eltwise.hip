// Fused per-channel bias + residual + ReLU for channels-last (NHWC) fp32 activations: the epilogue of
// every convolution of the ResNet-50 trunk once its frozen batch-norm is folded into the weights
// (detectinblur_amd/models/backbone.py).  Stock eager PyTorch runs it as 2-4 full passes over the
// activation (bias add, residual add, ReLU); here it is one pass, in place on the convolution output.
// Pure streaming: 8 bytes per element without a residual, 12 with one -> HBM-bound.
#include "dib_common.h"
#include <stdlib.h>

namespace dib {

// MASK: also write the ReLU's sign pattern, one byte per float4 (bits 0-3 = element > 0): what the backward pass needs of
// the output, at 1/16 of its size (the backward then reads 4.25 instead of 8 bytes per element).
template <bool RES, bool RELU, bool MASK>
__global__ __launch_bounds__(256) void bias_act_vec4_kernel(float4 *__restrict__ x, const float4 *__restrict__ bias,
                                                           const float4 *__restrict__ res, long long n4, int C4,
                                                           unsigned char *__restrict__ mask) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = x[i];
    const float4 b = bias[(int)(i % C4)];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (RES) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    x[i] = v;
    if (MASK) mask[i] = (unsigned char)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
  }
}

// ReLU backward from that mask: out = mask ? grad : 0 (torch's threshold_backward(grad, y, 0) with y > 0 read from the mask).
__global__ __launch_bounds__(256) void relu_mask_bwd_kernel(const float4 *__restrict__ g, const unsigned char *__restrict__ mask,
                                                           float4 *__restrict__ out, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = g[i];
    const unsigned m = mask[i];
    v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
    out[i] = v;
  }
}

// Gradient accumulation at a residual block's input fused with the ReLU backward of the tensor it belongs to:
// a = (a + b) [masked], one pass (12.25 B per element) where autograd's add followed by a mask pass moves 20.25.
template <bool MASK>
__global__ __launch_bounds__(256) void add_mask_kernel(float4 *__restrict__ a, const float4 *__restrict__ b,
                                                      const unsigned char *__restrict__ mask, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = a[i];
    const float4 w = b[i];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    if (MASK) {
      const unsigned m = mask[i];
      v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
    }
    a[i] = v;
  }
}

// Data gradient of a strided 1x1 convolution added into the data gradient of the stride-1 convolution that shares its input
// (the downsample path of a ResNet stage's first block): a[n, ys * s, xs * s, :] += b[n, ys, xs, :], channels-last, in place.
// Replaces a zero-filled full-size gradient plus a full-size add by a pass over a quarter of the pixels.
__global__ __launch_bounds__(256) void scatter_add_kernel(float4 *__restrict__ a, const float4 *__restrict__ b, int Hs, int Ws, int C4,
                                                         int H, int W, int s, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long p = i / C4;
    const int xs = (int)(p % Ws);
    p /= Ws;
    const int ys = (int)(p % Hs);
    const long long n = p / Hs;
    const long long j = ((n * H + (long long)ys * s) * W + (long long)xs * s) * C4 + c;
    float4 v = a[j];
    const float4 w = b[i];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    a[j] = v;
  }
}

// FPN top-down merge in one pass, in place on the lateral convolution's output:
//   x[n, h, w, :] += bias[:] + top[n, sh(h), sw(w), :]      (lateral + bias + interpolate(top, size=(H, W), mode="nearest"))
// with ATen's nearest source index, src = min(int(floorf(dst * float(in) / out)), in - 1).  Stock PyTorch runs it as a bias add,
// an upsample that writes a full-size tensor and an add that reads it back (7 tensor passes); this is 2.25.
__global__ __launch_bounds__(256) void topdown_merge_kernel(float4 *__restrict__ x, const float4 *__restrict__ bias,
                                                           const float4 *__restrict__ top, int H, int W, int Ht, int Wt, int C4,
                                                           float scale_h, float scale_w, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long p = i / C4;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const long long n = p / H;
    const int sh = min((int)floorf((float)h * scale_h), Ht - 1), sw = min((int)floorf((float)w * scale_w), Wt - 1);
    float4 v = x[i];
    const float4 b = bias[c];
    const float4 t = top[((n * Ht + sh) * Wt + sw) * C4 + c];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    x[i] = v;
  }
}

// ResNet stem: bias + ReLU + 3x3 / stride 2 / padding 1 max-pool of the first convolution's output in ONE pass.
// relu and max commute, so pooled = relu(max over the window of (x + bias)); the backward pass needs, per pooled element, only
// WHICH window position won (4 bits; 15 = the maximum was not positive, no gradient): the 550 MB activation is neither written
// back nor re-read, and ATen's int64 index tensor (2 x the pooled output) disappears.  Window scan order and the strict `>`
// are ATen's (max_pool2d: first maximum in row-major window order).
__global__ __launch_bounds__(256) void stem_pool_fwd_kernel(const float4 *__restrict__ x, const float4 *__restrict__ bias,
                                                           float4 *__restrict__ out, unsigned short *__restrict__ arg, int H, int W,
                                                           int Ho, int Wo, int C4) {
  // grid: x over the Wo * C4 float4 of one pooled row, y = pooled row, z = image (no 64-bit divisions on the way to an address)
  const unsigned col = blockIdx.x * 256u + threadIdx.x;
  if (col >= (unsigned)(Wo * C4)) return;
  const int ow = (int)(col / (unsigned)C4), c = (int)(col % (unsigned)C4), oh = blockIdx.y;
  const size_t n = blockIdx.z;
  const float4 b = bias[c];
  const float ninf = -__builtin_inff();
  float4 m = make_float4(ninf, ninf, ninf, ninf);
  unsigned ax = 15, ay = 15, az = 15, aw = 15;
  // all nine loads are issued before the first comparison (clamped addresses, out-of-range positions replaced by -inf): one
  // memory round trip per thread instead of up to nine dependent ones
  float4 v[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int h = oh * 2 - 1 + k / 3, w = ow * 2 - 1 + k % 3;
    const int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1);
    v[k] = x[((n * H + hc) * W + wc) * C4 + c];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int h = oh * 2 - 1 + k / 3, w = ow * 2 - 1 + k % 3;
    const bool in = h >= 0 && h < H && w >= 0 && w < W;
    const float vx = in ? v[k].x + b.x : ninf, vy = in ? v[k].y + b.y : ninf, vz = in ? v[k].z + b.z : ninf, vw = in ? v[k].w + b.w : ninf;
    if (vx > m.x) { m.x = vx; ax = k; }
    if (vy > m.y) { m.y = vy; ay = k; }
    if (vz > m.z) { m.z = vz; az = k; }
    if (vw > m.w) { m.w = vw; aw = k; }
  }
  if (!(m.x > 0.f)) { m.x = 0.f; ax = 15; }
  if (!(m.y > 0.f)) { m.y = 0.f; ay = 15; }
  if (!(m.z > 0.f)) { m.z = 0.f; az = 15; }
  if (!(m.w > 0.f)) { m.w = 0.f; aw = 15; }
  const size_t i = ((n * Ho + oh) * Wo) * C4 + col;
  out[i] = m;
  arg[i] = (unsigned short)(ax | (ay << 4) | (az << 8) | (aw << 12));
}

// Its backward: the gradient of the convolution output, dense (every input pixel belongs to at most 2 x 2 windows; it takes the
// pooled gradient of those whose recorded winner it is).  One pass: pooled gradient + 2 bytes per 4 pooled elements in,
// full-size gradient out -- instead of ATen's max-pool backward plus the ReLU mask pass.
__global__ __launch_bounds__(256) void stem_pool_bwd_kernel(const float4 *__restrict__ g_out, const unsigned short *__restrict__ arg,
                                                           float4 *__restrict__ g_in, int H, int W, int Ho, int Wo, int C4) {
  // grid: x over the W * C4 float4 of one input row, y = input row, z = image
  const unsigned col = blockIdx.x * 256u + threadIdx.x;
  if (col >= (unsigned)(W * C4)) return;
  const int w = (int)(col / (unsigned)C4), c = (int)(col % (unsigned)C4), h = blockIdx.y;
  const size_t n = blockIdx.z;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  // the (at most) 2 x 2 windows that contain (h, w): oh in {h >> 1, (h + 1) >> 1}, likewise ow; all loads first, then the
  // selection -- one memory round trip per thread
  const int ohs[2] = {h >> 1, (h + 1) >> 1}, ows[2] = {w >> 1, (w + 1) >> 1};
  unsigned a[4];
  float4 go[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int oh = min(ohs[q >> 1], Ho - 1), ow = min(ows[q & 1], Wo - 1);
    const size_t j = ((n * Ho + oh) * Wo + ow) * C4 + c;
    a[q] = arg[j];
    go[q] = g_out[j];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int oh = ohs[q >> 1], ow = ows[q & 1];
    // a window counts once: the second candidate equals the first for even h (w), and must lie inside the pooled image
    const bool use = oh < Ho && ow < Wo && ((q >> 1) == 0 || ohs[1] != ohs[0]) && ((q & 1) == 0 || ows[1] != ows[0]);
    const unsigned k = use ? (unsigned)(h - (oh * 2 - 1)) * 3u + (unsigned)(w - (ow * 2 - 1)) : 14u;      // 14: never recorded
    if ((a[q] & 15u) == k) g.x += go[q].x;
    if (((a[q] >> 4) & 15u) == k) g.y += go[q].y;
    if (((a[q] >> 8) & 15u) == k) g.z += go[q].z;
    if ((a[q] >> 12) == k) g.w += go[q].w;
  }
  g_in[((n * H + h) * W) * C4 + col] = g;
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bias_act_scalar_kernel(float *__restrict__ x, const float *__restrict__ bias,
                                                             const float *__restrict__ res, long long n, int C) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = x[i] + bias[(int)(i % C)];
    if (RES) v += res[i];
    if (RELU) v = fmaxf(v, 0.f);
    x[i] = v;
  }
}


// ---- frozen batch-norm folds of many convolutions in ONE launch --------------------------------------------------------------
// Training folds every trunk convolution's frozen batch-norm into its weight each step (the weights move): per convolution
// that is five tiny launches forward (add eps, rsqrt, two multiplies, a subtract, the weight multiply) and one backward, 53
// times -- ~320 launches of ~5 us that the GPU waits for (profiles/r4_train_step_conv.txt).  Here: up to FOLD_MAX pairs per launch.
// Same operations in the same order as the tensor expressions (scale = bn_w * rsqrt(var + eps); shift = bn_b - mean * scale;
// wf = w * scale[co]), so the results are bit-identical to the per-convolution path (tests/test_detector_ops.py).
constexpr int FOLD_MAX = 32;
struct FoldArgs {
  const float *w[FOLD_MAX], *bn_w[FOLD_MAX], *bn_b[FOLD_MAX], *mean[FOLD_MAX], *var[FOLD_MAX];
  float *wf[FOLD_MAX], *scale[FOLD_MAX], *shift[FOLD_MAX];
  int inner[FOLD_MAX];            // elements per output channel (Ci * kh * kw)
  long long count[FOLD_MAX];      // Co * inner
  int block_start[FOLD_MAX + 1];  // first 1024-element block of pair k
  int n;
  float eps;
};

// pass 1: one thread per output channel.  torch's rsqrt kernel on this platform returns the correctly rounded value (checked
// against float(rsqrt(double(x))) on 200,000 samples, scratch/t_rsq.py) where rsqrtf() is v_rsq_f32's 1-ulp approximation
// (12 % of the samples differ): the double-precision form below reproduces torch's bits.
__global__ __launch_bounds__(256) void fold_scale_shift_kernel(FoldArgs a) {
#pragma clang fp contract(off)
  const int k = blockIdx.y;
  const int co = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.n || (long long)co * a.inner[k] >= a.count[k]) return;
  const float v = a.var[k][co] + a.eps;
  const float sc = a.bn_w[k][co] * (float)(1.0 / sqrt((double)v));
  a.scale[k][co] = sc;
  a.shift[k][co] = a.bn_b[k][co] - a.mean[k][co] * sc;
}

// pass 2: wf = w * scale[co], 1024 consecutive elements of one weight per block
__global__ __launch_bounds__(256) void fold_bn_multi_kernel(FoldArgs a) {
#pragma clang fp contract(off)
  int k = 0;
  while (k + 1 < a.n && (int)blockIdx.x >= a.block_start[k + 1]) ++k;
  const long long base = (long long)((int)blockIdx.x - a.block_start[k]) * 1024;
  const int inner = a.inner[k];
  const float *w = a.w[k];
  const float *sc = a.scale[k];
  float *wf = a.wf[k];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long e = base + j * 256 + threadIdx.x;
    if (e >= a.count[k]) break;
    wf[e] = w[e] * sc[(int)(e / inner)];
  }
}

struct ScaleRowsArgs {
  const float *g[FOLD_MAX], *scale[FOLD_MAX];
  float *dw[FOLD_MAX];
  int inner[FOLD_MAX];
  long long count[FOLD_MAX];
  int block_start[FOLD_MAX + 1];
  int n;
};

__global__ __launch_bounds__(256) void scale_rows_multi_kernel(ScaleRowsArgs a) {
  int k = 0;
  while (k + 1 < a.n && (int)blockIdx.x >= a.block_start[k + 1]) ++k;
  const long long base = (long long)((int)blockIdx.x - a.block_start[k]) * 1024;
  const int inner = a.inner[k];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long e = base + j * 256 + threadIdx.x;
    if (e >= a.count[k]) break;
    a.dw[k][e] = a.g[k][e] * a.scale[k][(int)(e / inner)];
  }
}

}  // namespace dib

using namespace dib;

// Launch shape of the streaming kernels above: ONE float4 per thread (the grid-stride loops only matter past 2^30 workgroups).
// Measured on the 550 MB tensors of the detector's first pyramid level (scratch/ubench/ub_stream.hip): 185 us (5.9 TB/s) against
// 220 us (5.0 TB/s) with the grid capped at 32 workgroups per CU, 197 against 245 us with the sign mask.
static long long max_blocks_from_env() {       // DIB_ELTWISE_MAX_BLOCKS: A/B knob for that measurement (8192 = the old cap)
  const char *e = getenv("DIB_ELTWISE_MAX_BLOCKS");
  const long long v = e ? atoll(e) : 0;
  return v > 0 ? v : (1ll << 30);
}
static const long long MAX_BLOCKS = max_blocks_from_env();

// x_dev: [n_elems] fp32 viewed as [..., C] with the channel fastest (NHWC storage), updated in place:
//   x = act(x + bias[c] (+ residual)),  act = ReLU when relu != 0.
static int bias_act_impl(float *x_dev, const float *bias_dev, const float *residual_dev, long long n_elems, int C, int relu,
                         unsigned char *mask_dev, void *stream);

extern "C" int dib_bias_act_nhwc(float *x_dev, const float *bias_dev, const float *residual_dev, long long n_elems, int C,
                                 int relu, void *stream) {
  return bias_act_impl(x_dev, bias_dev, residual_dev, n_elems, C, relu, nullptr, stream);
}

// Same, and mask_dev[n_elems / 4] receives the sign pattern of the result (one byte per 4 consecutive elements, bit k =
// element 4i + k > 0): dib_relu_mask_backward's input.  Needs C % 4 == 0 and 16-byte aligned pointers.
extern "C" int dib_bias_act_mask_nhwc(float *x_dev, const float *bias_dev, const float *residual_dev, long long n_elems, int C,
                                      unsigned char *mask_dev, void *stream) {
  if (!mask_dev) { set_error("dib_bias_act_mask_nhwc: null mask pointer"); return DIB_EINVAL; }
  if ((C % 4) != 0 || (((uintptr_t)x_dev | (uintptr_t)bias_dev | (uintptr_t)residual_dev) & 15) != 0) {
    set_error("dib_bias_act_mask_nhwc: needs C %% 4 == 0 and 16-byte aligned tensors");
    return DIB_EINVAL;
  }
  return bias_act_impl(x_dev, bias_dev, residual_dev, n_elems, C, 1, mask_dev, stream);
}

// grad_out = mask ? grad_in : 0 over n_elems fp32 values (n_elems % 4 == 0, 16-byte aligned; grad_out may alias grad_in).
extern "C" int dib_relu_mask_backward(const float *grad_in_dev, const unsigned char *mask_dev, float *grad_out_dev, long long n_elems,
                                      void *stream) {
  if (n_elems < 0 || (n_elems % 4) != 0) { set_error("dib_relu_mask_backward: n_elems must be a non-negative multiple of 4"); return DIB_EINVAL; }
  if (n_elems == 0) return DIB_OK;
  if (!grad_in_dev || !mask_dev || !grad_out_dev) { set_error("dib_relu_mask_backward: null pointer"); return DIB_EINVAL; }
  if ((((uintptr_t)grad_in_dev | (uintptr_t)grad_out_dev) & 15) != 0) { set_error("dib_relu_mask_backward: tensors must be 16-byte aligned"); return DIB_EINVAL; }
  const long long n4 = n_elems / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)grad_in_dev, mask_dev,
                     (float4 *)grad_out_dev, n4);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

// a = (a + b), then zeroed where the mask bit is clear (mask_dev NULL: plain accumulate).  n_elems % 4 == 0, 16-byte aligned.
extern "C" int dib_add_relu_mask(float *a_dev, const float *b_dev, const unsigned char *mask_dev, long long n_elems, void *stream) {
  if (n_elems < 0 || (n_elems % 4) != 0) { set_error("dib_add_relu_mask: n_elems must be a non-negative multiple of 4"); return DIB_EINVAL; }
  if (n_elems == 0) return DIB_OK;
  if (!a_dev || !b_dev) { set_error("dib_add_relu_mask: null pointer"); return DIB_EINVAL; }
  if ((((uintptr_t)a_dev | (uintptr_t)b_dev) & 15) != 0) { set_error("dib_add_relu_mask: tensors must be 16-byte aligned"); return DIB_EINVAL; }
  const long long n4 = n_elems / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  if (mask_dev) hipLaunchKernelGGL(add_mask_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)a_dev, (const float4 *)b_dev, mask_dev, n4);
  else hipLaunchKernelGGL(add_mask_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)a_dev, (const float4 *)b_dev, (const unsigned char *)nullptr, n4);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

// a[N, H, W, C] (channels-last fp32) += b[N, Hs, Ws, C] at the pixels (ys * stride, xs * stride); C % 4 == 0, 16-byte aligned.
extern "C" int dib_scatter_add_nhwc(float *a_dev, const float *b_dev, int N, int H, int W, int Hs, int Ws, int C, int stride, void *stream) {
  if (N < 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || (C % 4) != 0 || stride < 1) { set_error("dib_scatter_add_nhwc: bad shape (C %% 4 == 0)"); return DIB_EINVAL; }
  if ((long long)(Hs - 1) * stride > H - 1 || (long long)(Ws - 1) * stride > W - 1) { set_error("dib_scatter_add_nhwc: strided grid leaves the target"); return DIB_ESHAPE; }
  if (N == 0) return DIB_OK;
  if (!a_dev || !b_dev) { set_error("dib_scatter_add_nhwc: null pointer"); return DIB_EINVAL; }
  if ((((uintptr_t)a_dev | (uintptr_t)b_dev) & 15) != 0) { set_error("dib_scatter_add_nhwc: tensors must be 16-byte aligned"); return DIB_EINVAL; }
  const long long n4 = (long long)N * Hs * Ws * (C / 4);
  long long blocks = (n4 + 255) / 256;
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  hipLaunchKernelGGL(scatter_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)a_dev, (const float4 *)b_dev, Hs, Ws,
                     C / 4, H, W, stride, n4);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

// x[N, H, W, C] (channels-last fp32, in place) += bias[C] + top[N, Ht, Wt, C] at the nearest-neighbour source pixel;
// C % 4 == 0, 16-byte aligned.
extern "C" int dib_fpn_topdown_merge_nhwc(float *x_dev, const float *bias_dev, const float *top_dev, int N, int H, int W, int Ht, int Wt,
                                          int C, void *stream) {
  if (N < 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0 || C <= 0 || (C % 4) != 0) { set_error("dib_fpn_topdown_merge_nhwc: bad shape (C %% 4 == 0)"); return DIB_EINVAL; }
  if (N == 0) return DIB_OK;
  if (!x_dev || !bias_dev || !top_dev) { set_error("dib_fpn_topdown_merge_nhwc: null pointer"); return DIB_EINVAL; }
  if ((((uintptr_t)x_dev | (uintptr_t)bias_dev | (uintptr_t)top_dev) & 15) != 0) { set_error("dib_fpn_topdown_merge_nhwc: tensors must be 16-byte aligned"); return DIB_EINVAL; }
  const long long n4 = (long long)N * H * W * (C / 4);
  long long blocks = (n4 + 255) / 256;
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  hipLaunchKernelGGL(topdown_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)x_dev, (const float4 *)bias_dev,
                     (const float4 *)top_dev, H, W, Ht, Wt, C / 4, (float)Ht / (float)H, (float)Wt / (float)W, n4);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

// out[N, Ho, Wo, C] = max_pool2d(relu(x[N, H, W, C] + bias[C]), 3, stride 2, padding 1), Ho = (H - 1) / 2 + 1; arg: one
// unsigned short per 4 output channels (4-bit window position of the winner per channel, 15 = no gradient).  C % 4 == 0.
extern "C" int dib_stem_pool_forward(const float *x_dev, const float *bias_dev, float *out_dev, unsigned short *arg_dev, int N, int H, int W,
                                     int C, void *stream) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) != 0) { set_error("dib_stem_pool_forward: bad shape (C %% 4 == 0)"); return DIB_EINVAL; }
  if (N == 0) return DIB_OK;
  if (!x_dev || !bias_dev || !out_dev || !arg_dev) { set_error("dib_stem_pool_forward: null pointer"); return DIB_EINVAL; }
  if ((((uintptr_t)x_dev | (uintptr_t)bias_dev | (uintptr_t)out_dev) & 15) != 0) { set_error("dib_stem_pool_forward: tensors must be 16-byte aligned"); return DIB_EINVAL; }
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (Ho > 65535 || N > 65535) { set_error("dib_stem_pool_forward: at most 65535 pooled rows and images per call"); return DIB_ESHAPE; }
  hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3((unsigned)((Wo * (C / 4) + 255) / 256), (unsigned)Ho, (unsigned)N), dim3(256), 0, (hipStream_t)stream,
                     (const float4 *)x_dev, (const float4 *)bias_dev, (float4 *)out_dev, arg_dev, H, W, Ho, Wo, C / 4);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

// grad_in[N, H, W, C] from grad_out[N, Ho, Wo, C] and the forward pass's arg.
extern "C" int dib_stem_pool_backward(const float *grad_out_dev, const unsigned short *arg_dev, float *grad_in_dev, int N, int H, int W, int C,
                                      void *stream) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) != 0) { set_error("dib_stem_pool_backward: bad shape (C %% 4 == 0)"); return DIB_EINVAL; }
  if (N == 0) return DIB_OK;
  if (!grad_out_dev || !arg_dev || !grad_in_dev) { set_error("dib_stem_pool_backward: null pointer"); return DIB_EINVAL; }
  if ((((uintptr_t)grad_out_dev | (uintptr_t)grad_in_dev) & 15) != 0) { set_error("dib_stem_pool_backward: tensors must be 16-byte aligned"); return DIB_EINVAL; }
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (H > 65535 || N > 65535) { set_error("dib_stem_pool_backward: at most 65535 rows and images per call"); return DIB_ESHAPE; }
  hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3((unsigned)((W * (C / 4) + 255) / 256), (unsigned)H, (unsigned)N), dim3(256), 0, (hipStream_t)stream,
                     (const float4 *)grad_out_dev, arg_dev, (float4 *)grad_in_dev, H, W, Ho, Wo, C / 4);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

static int bias_act_impl(float *x_dev, const float *bias_dev, const float *residual_dev, long long n_elems, int C, int relu,
                         unsigned char *mask_dev, void *stream) {
  if (n_elems < 0 || C <= 0 || (n_elems % C) != 0) { set_error("dib_bias_act_nhwc: n_elems must be a multiple of C"); return DIB_EINVAL; }
  if (n_elems == 0) return DIB_OK;
  if (!x_dev || !bias_dev) { set_error("dib_bias_act_nhwc: null pointer"); return DIB_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (C % 4 == 0) && (((uintptr_t)x_dev | (uintptr_t)bias_dev | (uintptr_t)residual_dev) & 15) == 0;
  const long long n = vec ? n_elems / 4 : n_elems;
  long long blocks = (n + 255) / 256;
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
#define DIB_LAUNCH(RES, RELU)                                                                                         \
  do {                                                                                                                \
    if (vec && mask_dev) hipLaunchKernelGGL((bias_act_vec4_kernel<RES, true, true>), dim3((unsigned)blocks), dim3(256), 0, s,        \
                                (float4 *)x_dev, (const float4 *)bias_dev, (const float4 *)residual_dev, n, C / 4, mask_dev); \
    else if (vec) hipLaunchKernelGGL((bias_act_vec4_kernel<RES, RELU, false>), dim3((unsigned)blocks), dim3(256), 0, s, (float4 *)x_dev, \
                                (const float4 *)bias_dev, (const float4 *)residual_dev, n, C / 4, (unsigned char *)nullptr); \
    else hipLaunchKernelGGL((bias_act_scalar_kernel<RES, RELU>), dim3((unsigned)blocks), dim3(256), 0, s, x_dev, bias_dev,   \
                            residual_dev, n, C);                                                                      \
  } while (0)
  if (residual_dev) { if (relu) DIB_LAUNCH(true, true); else DIB_LAUNCH(true, false); }
  else { if (relu) DIB_LAUNCH(false, true); else DIB_LAUNCH(false, false); }
#undef DIB_LAUNCH
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

// Frozen batch-norm folds of n convolutions (any n; FOLD_MAX pairs per launch).  Host arrays of n device pointers / sizes:
// w[k] [Co[k]][inner[k]] dense with the output channel outermost (contiguous OR channels-last weights), bn_* [Co[k]];
// wf[k] receives the folded weight in w[k]'s element order, scale[k] / shift[k] [Co[k]].
extern "C" int dib_fold_bn_multi(const float *const *w, const float *const *bn_w, const float *const *bn_b, const float *const *mean,
                                 const float *const *var, const int *Co, const int *inner, int n, float eps, float *const *wf,
                                 float *const *scale, float *const *shift, void *stream) {
  if (n < 0 || (n > 0 && (!w || !bn_w || !bn_b || !mean || !var || !Co || !inner || !wf || !scale || !shift))) { set_error("dib_fold_bn_multi: null pointer or negative count"); return DIB_EINVAL; }
  for (int k0 = 0; k0 < n; k0 += FOLD_MAX) {
    FoldArgs a;
    a.n = n - k0 < FOLD_MAX ? n - k0 : FOLD_MAX;
    a.eps = eps;
    int blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      const int k = k0 + i;
      if (!w[k] || !bn_w[k] || !bn_b[k] || !mean[k] || !var[k] || !wf[k] || !scale[k] || !shift[k] || Co[k] <= 0 || inner[k] <= 0) { set_error("dib_fold_bn_multi: pair %d has a null pointer or an empty shape", k); return DIB_EINVAL; }
      a.w[i] = w[k]; a.bn_w[i] = bn_w[k]; a.bn_b[i] = bn_b[k]; a.mean[i] = mean[k]; a.var[i] = var[k];
      a.wf[i] = wf[k]; a.scale[i] = scale[k]; a.shift[i] = shift[k];
      a.inner[i] = inner[k]; a.count[i] = (long long)Co[k] * inner[k];
      a.block_start[i] = blocks;
      blocks += (int)((a.count[i] + 1023) / 1024);
    }
    for (int i = a.n; i <= FOLD_MAX; ++i) a.block_start[i] = blocks;
    int max_co = 0;
    for (int i = 0; i < a.n; ++i) max_co = Co[k0 + i] > max_co ? Co[k0 + i] : max_co;
    hipLaunchKernelGGL(fold_scale_shift_kernel, dim3((unsigned)((max_co + 255) / 256), (unsigned)a.n), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(fold_bn_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  }
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

// The backward of those folds: dw[k][co][...] = g[k][co][...] * scale[k][co] (the gradient of w * scale[co] with respect to w).
extern "C" int dib_scale_rows_multi(const float *const *g, const float *const *scale, const int *Co, const int *inner, int n,
                                    float *const *dw, void *stream) {
  if (n < 0 || (n > 0 && (!g || !scale || !Co || !inner || !dw))) { set_error("dib_scale_rows_multi: null pointer or negative count"); return DIB_EINVAL; }
  for (int k0 = 0; k0 < n; k0 += FOLD_MAX) {
    ScaleRowsArgs a;
    a.n = n - k0 < FOLD_MAX ? n - k0 : FOLD_MAX;
    int blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      const int k = k0 + i;
      if (!g[k] || !scale[k] || !dw[k] || Co[k] <= 0 || inner[k] <= 0) { set_error("dib_scale_rows_multi: tensor %d has a null pointer or an empty shape", k); return DIB_EINVAL; }
      a.g[i] = g[k]; a.scale[i] = scale[k]; a.dw[i] = dw[k];
      a.inner[i] = inner[k]; a.count[i] = (long long)Co[k] * inner[k];
      a.block_start[i] = blocks;
      blocks += (int)((a.count[i] + 1023) / 1024);
    }
    for (int i = a.n; i <= FOLD_MAX; ++i) a.block_start[i] = blocks;
    hipLaunchKernelGGL(scale_rows_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  }
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

// ---- bias (+ ReLU) with the layout change MIOpen's planar kernels need, in one pass ------------------------------------------------
// At batch 1 the wide 3x3 convolutions of ResNet layer2-4 run 1.4-1.7x faster through MIOpen's planar (NCHW) kernels than through
// its channels-last ones (models/backbone.py: _as_planar).  Around each of them eager PyTorch spent four passes: the epilogue of the
// convolution before (in place, channels-last), a copy to planar, a copy of the result back to channels-last, its epilogue
// (profiles/r4_trunk_b1_trace.txt: 0.68 ms of copies per image, the strided NHWC -> NCHW copy at 0.6 TB/s).  These are the two
// middle pairs as one pass each: a 64 x 64 tile transpose through LDS with the bias indexed on the channel axis.
//   to_planar = 1: in [N][HW][C] -> out [N][C][HW];   to_planar = 0: in [N][C][HW] -> out [N][HW][C].   out = act(in + bias[c])
namespace dib {

template <bool RELU>
__global__ __launch_bounds__(256) void bias_act_transpose_kernel(const float *__restrict__ in, const float *__restrict__ bias, float *__restrict__ out,
                                                                int rows, int cols, int bias_on_cols) {
  // in: [rows][cols] per image, out: [cols][rows]
  __shared__ float tile[64][65];
  const size_t img = (size_t)blockIdx.z * rows * cols;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
  const int c = c0 + tx;
  const float bc = (bias_on_cols && c < cols) ? bias[c] : 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int r = r0 + ty + 4 * k;
    if (r < rows && c < cols) {
      float v = in[img + (size_t)r * cols + c] + (bias_on_cols ? bc : bias[r]);
      if (RELU) v = fmaxf(v, 0.f);                               // as bias_act_vec4_kernel
      tile[ty + 4 * k][tx] = v;
    }
  }
  __syncthreads();
  const int orow = r0 + tx;                                     // output: [cols][rows]: consecutive lanes along `rows`
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int oc = c0 + ty + 4 * k;
    if (oc < cols && orow < rows) out[img + (size_t)oc * rows + orow] = tile[tx][ty + 4 * k];
  }
}

}  // namespace dib

extern "C" int dib_bias_act_transpose(const float *in_dev, const float *bias_dev, float *out_dev, int N, int C, long long HW, int to_planar, int relu,
                                      void *stream) {
  if (N < 0 || C <= 0 || HW <= 0 || HW > 0x7fffffffLL) { set_error("dib_bias_act_transpose: bad shape"); return DIB_EINVAL; }
  if (N == 0) return DIB_OK;
  if (!in_dev || !bias_dev || !out_dev || in_dev == out_dev) { set_error("dib_bias_act_transpose: null or aliased pointers"); return DIB_EINVAL; }
  const int rows = to_planar ? (int)HW : C, cols = to_planar ? C : (int)HW;
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64, N);
  if (grid.y > 65535 || grid.z > 65535) { set_error("dib_bias_act_transpose: tensor too large for one launch"); return DIB_EINVAL; }
  if (relu) hipLaunchKernelGGL(dib::bias_act_transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in_dev, bias_dev, out_dev, rows, cols, to_planar);
  else hipLaunchKernelGGL(dib::bias_act_transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in_dev, bias_dev, out_dev, rows, cols, to_planar);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
