// Detector ops that stock PyTorch-ROCm does not provide (torchvision is not a dependency):
// RoIAlign forward / backward and NMS, fp32, for the Faster R-CNN heads that the blurred batches
// feed (reference models/faster_rcnn.py:204-208 uses torchvision.ops.MultiScaleRoIAlign(7, sr=2);
// RegionProposalNetwork / RoIHeads call torchvision.ops.nms).  Semantics follow the published
// Detectron / Mask R-CNN definition of RoIAlign with aligned=False (no half-pixel shift, RoI
// width/height clamped to >= 1) so that results match a plain-PyTorch fp32 restatement
// (tests/test_detector_ops.py).
#include "dib_common.h"

namespace dib {

// bilinear sample of one H x W plane at (y, x); out-of-range samples contribute 0 (Detectron rule)
__device__ inline float bilinear(const float *__restrict__ p, int H, int W, float y, float x) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
  y = fmaxf(y, 0.f);
  x = fmaxf(x, 0.f);
  int y0 = (int)y, x0 = (int)x, y1, x1;
  if (y0 >= H - 1) { y0 = y1 = H - 1; y = (float)y0; } else y1 = y0 + 1;
  if (x0 >= W - 1) { x0 = x1 = W - 1; x = (float)x0; } else x1 = x0 + 1;
  const float ly = y - y0, lx = x - x0, hy = 1.f - ly, hx = 1.f - lx;
  return hy * hx * p[y0 * W + x0] + hy * lx * p[y0 * W + x1] + ly * hx * p[y1 * W + x0] + ly * lx * p[y1 * W + x1];
}

// one thread per output element; consecutive threads walk pw, ph, c of one RoI: neighbouring
// lanes sample neighbouring feature pixels (coalesced within a bin row)
__global__ void roi_align_fwd_kernel(const float *__restrict__ feat, const float *__restrict__ rois, int K, int C, int H,
                                     int W, float scale, int P, int sr, int aligned, float *__restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)K * C * P * P;
  if (idx >= total) return;
  const int pw = (int)(idx % P), ph = (int)((idx / P) % P), c = (int)((idx / ((long long)P * P)) % C);
  const int k = (int)(idx / ((long long)P * P * C));
  const float *r = rois + (size_t)k * 5;
  const int b = (int)r[0];
  const float off = aligned ? 0.5f : 0.f;
  const float x1 = r[1] * scale - off, y1 = r[2] * scale - off, x2 = r[3] * scale - off, y2 = r[4] * scale - off;
  float rw = x2 - x1, rh = y2 - y1;
  if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  const float bh = rh / P, bw = rw / P;
  const int gh = sr > 0 ? sr : (int)ceilf(rh / P), gw = sr > 0 ? sr : (int)ceilf(rw / P);
  const float cnt = (float)max(gh * gw, 1);
  const float *plane = feat + ((size_t)b * C + c) * H * W;
  float acc = 0.f;
  for (int iy = 0; iy < gh; ++iy) {
    const float y = y1 + ph * bh + (iy + 0.5f) * bh / gh;
    for (int ix = 0; ix < gw; ++ix) {
      const float x = x1 + pw * bw + (ix + 0.5f) * bw / gw;
      acc += bilinear(plane, H, W, y, x);
    }
  }
  out[idx] = acc / cnt;
}

__global__ void roi_align_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ rois, int K, int C, int H,
                                     int W, float scale, int P, int sr, int aligned, float *__restrict__ gfeat) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)K * C * P * P;
  if (idx >= total) return;
  const int pw = (int)(idx % P), ph = (int)((idx / P) % P), c = (int)((idx / ((long long)P * P)) % C);
  const int k = (int)(idx / ((long long)P * P * C));
  const float *r = rois + (size_t)k * 5;
  const int b = (int)r[0];
  const float off = aligned ? 0.5f : 0.f;
  const float x1 = r[1] * scale - off, y1 = r[2] * scale - off, x2 = r[3] * scale - off, y2 = r[4] * scale - off;
  float rw = x2 - x1, rh = y2 - y1;
  if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  const float bh = rh / P, bw = rw / P;
  const int gh = sr > 0 ? sr : (int)ceilf(rh / P), gw = sr > 0 ? sr : (int)ceilf(rw / P);
  const float g = gout[idx] / (float)max(gh * gw, 1);
  float *plane = gfeat + ((size_t)b * C + c) * H * W;
  for (int iy = 0; iy < gh; ++iy) {
    float y = y1 + ph * bh + (iy + 0.5f) * bh / gh;
    for (int ix = 0; ix < gw; ++ix) {
      float x = x1 + pw * bw + (ix + 0.5f) * bw / gw;
      if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
      float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
      int y0 = (int)yy, x0 = (int)xx, y1i, x1i;
      if (y0 >= H - 1) { y0 = y1i = H - 1; yy = (float)y0; } else y1i = y0 + 1;
      if (x0 >= W - 1) { x0 = x1i = W - 1; xx = (float)x0; } else x1i = x0 + 1;
      const float ly = yy - y0, lx = xx - x0, hy = 1.f - ly, hx = 1.f - lx;
      atomicAdd(plane + y0 * W + x0, g * hy * hx);
      atomicAdd(plane + y0 * W + x1i, g * hy * lx);
      atomicAdd(plane + y1i * W + x0, g * ly * hx);
      atomicAdd(plane + y1i * W + x1i, g * ly * lx);
    }
  }
}

// ---- NMS ----------------------------------------------------------------------------------------
// Pass 1: 64 x 64 blocks of the upper-triangular suppression matrix as 64-bit masks (one wave per
// block, boxes of the column block staged in LDS).  Pass 2: one wave walks the boxes in score order,
// lane w owning word w of the "removed" bit set; writes the kept indices and their count.
// Boxes must already be sorted by descending score.
__device__ inline float iou(const float4 a, const float4 b) {
  const float iw = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f), ih = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float inter = iw * ih, ua = (a.z - a.x) * (a.w - a.y) + (b.z - b.x) * (b.w - b.y) - inter;
  return inter / ua;
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float4 *__restrict__ boxes, int n, float thr,
                                                      unsigned long long *__restrict__ mask) {
  const int rb = blockIdx.y, cb = blockIdx.x, lane = threadIdx.x;
  if (cb < rb) return;  // only j > i matters
  const int nblk = (n + 63) / 64;
  __shared__ float4 cols[64];
  const int cj = cb * 64 + lane;
  if (cj < n) cols[lane] = boxes[cj];
  __syncthreads();
  const int i = rb * 64 + lane;
  if (i >= n) return;
  const float4 bi = boxes[i];
  unsigned long long m = 0;
  const int lim = min(64, n - cb * 64);
  for (int j = (rb == cb) ? lane + 1 : 0; j < lim; ++j)
    if (iou(bi, cols[j]) > thr) m |= 1ull << j;
  mask[(size_t)i * nblk + cb] = m;
}

__global__ __launch_bounds__(64) void nms_reduce_kernel(const unsigned long long *__restrict__ mask, int n,
                                                        long long *__restrict__ keep, int *__restrict__ count) {
  const int nblk = (n + 63) / 64, lane = threadIdx.x;
  // lane w holds removed-words w, w+64, ... (n <= 64*64*RW boxes)
  constexpr int RW = 4;  // up to 16384 boxes
  unsigned long long removed[RW] = {0, 0, 0, 0};
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    const int word = i >> 6, owner = word & 63, slot = word >> 6;
    unsigned long long w = 0;
#pragma unroll
    for (int s = 0; s < RW; ++s) if (s == slot) w = removed[s];
    const unsigned long long wi = __shfl(w, owner, 64);
    if (!((wi >> (i & 63)) & 1ull)) {
      if (lane == 0) keep[kept] = i;
      ++kept;
#pragma unroll
      for (int s = 0; s < RW; ++s) {
        const int wcol = lane + 64 * s;
        if (wcol < nblk && wcol >= word) removed[s] |= mask[(size_t)i * nblk + wcol];
      }
    }
  }
  if (lane == 0) *count = kept;
}

}  // namespace dib

using namespace dib;

extern "C" int dib_roi_align_forward(const float *feat_dev, const float *rois_dev, int K, int C, int H, int W,
                                     float spatial_scale, int pooled, int sampling_ratio, int aligned, float *out_dev,
                                     void *stream) {
  if (K < 0 || C <= 0 || H <= 0 || W <= 0 || pooled <= 0) { set_error("dib_roi_align_forward: bad shape"); return DIB_EINVAL; }
  if (K == 0) return DIB_OK;
  if (!feat_dev || !rois_dev || !out_dev) { set_error("dib_roi_align_forward: null pointer"); return DIB_EINVAL; }
  const long long total = (long long)K * C * pooled * pooled;
  hipLaunchKernelGGL(roi_align_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feat_dev,
                     rois_dev, K, C, H, W, spatial_scale, pooled, sampling_ratio, aligned, out_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" int dib_roi_align_backward(const float *grad_out_dev, const float *rois_dev, int K, int C, int H, int W,
                                      float spatial_scale, int pooled, int sampling_ratio, int aligned,
                                      float *grad_feat_dev, void *stream) {
  if (K < 0 || C <= 0 || H <= 0 || W <= 0 || pooled <= 0) { set_error("dib_roi_align_backward: bad shape"); return DIB_EINVAL; }
  if (K == 0) return DIB_OK;
  if (!grad_out_dev || !rois_dev || !grad_feat_dev) { set_error("dib_roi_align_backward: null pointer"); return DIB_EINVAL; }
  const long long total = (long long)K * C * pooled * pooled;
  hipLaunchKernelGGL(roi_align_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     grad_out_dev, rois_dev, K, C, H, W, spatial_scale, pooled, sampling_ratio, aligned, grad_feat_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" size_t dib_nms_workspace_bytes(int n) {
  if (n <= 0) return 0;
  const size_t nblk = ((size_t)n + 63) / 64;
  return (size_t)n * nblk * sizeof(unsigned long long);
}

extern "C" int dib_nms(const float *boxes_sorted_dev, int n, float iou_threshold, void *workspace_dev, long long *keep_dev,
                       int *count_dev, void *stream) {
  if (n < 0 || n > 16384) { set_error("dib_nms: n must be in [0, 16384], got %d", n); return DIB_EINVAL; }
  if (!count_dev) { set_error("dib_nms: null count pointer"); return DIB_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) { DIB_HIP_CHECK(hipMemsetAsync(count_dev, 0, sizeof(int), s)); return DIB_OK; }
  if (!boxes_sorted_dev || !workspace_dev || !keep_dev) { set_error("dib_nms: null pointer"); return DIB_EINVAL; }
  if (((uintptr_t)boxes_sorted_dev & 15) != 0) { set_error("dib_nms: boxes must be 16-byte aligned"); return DIB_EINVAL; }
  const int nblk = (n + 63) / 64;
  DIB_HIP_CHECK(hipMemsetAsync(workspace_dev, 0, dib_nms_workspace_bytes(n), s));
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nblk, nblk), dim3(64), 0, s, (const float4 *)boxes_sorted_dev, n, iou_threshold,
                     (unsigned long long *)workspace_dev);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(64), 0, s, (const unsigned long long *)workspace_dev, n, keep_dev, count_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
