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

// Measured alternative (MI355X, K=1024, C=256): lane = channel (64 atomics of one instruction on 64
// different planes) is 3.5x SLOWER (10.3 ms vs 2.9 ms): the L2 executes atomics per cache line, and
// the bin-major mapping above lets neighbouring lanes share lines.

// ---- channels-last (NHWC) RoIAlign over up to 4 pyramid levels in one launch ----------------------
// Workgroup = (RoI k, chunk of <= 256 channels); wave = one output bin at a time; lane = channel.
// In NHWC the C channels of a feature pixel are contiguous, so every load / atomic instruction of a
// wave covers 64 consecutive floats (two cache lines): the L2 executes atomics per line, and this is
// ~9x fewer line operations than the planar kernel above.  The [C][P*P] tile of the RoI goes through
// LDS so that the NCHW-flattened `out` / `grad_out` (what the box head's fc6 expects) is read and
// written coalesced.  Arithmetic order per element is that of the planar kernels (bit-identical).
struct RoiLevels {
  const float *feat[4];
  float *gfeat[4];
  int H[4], W[4];
  float scale[4];
  int n;
};
constexpr int ROI_CCHUNK = 256;

struct RoiGeom { float x1, y1, bw, bh; int gh, gw, b; float cnt; };

__device__ inline RoiGeom roi_geom(const float *r, float scale, int P, int sr, int aligned) {
  RoiGeom g;
  g.b = (int)r[0];
  const float off = aligned ? 0.5f : 0.f;
  g.x1 = r[1] * scale - off; g.y1 = r[2] * scale - off;
  const float x2 = r[3] * scale - off, y2 = r[4] * scale - off;
  float rw = x2 - g.x1, rh = y2 - g.y1;
  if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  g.bh = rh / P; g.bw = rw / P;
  g.gh = sr > 0 ? sr : (int)ceilf(rh / P); g.gw = sr > 0 ? sr : (int)ceilf(rw / P);
  g.cnt = (float)max(g.gh * g.gw, 1);
  return g;
}

__global__ __launch_bounds__(256) void roi_align_fwd_nhwc_kernel(RoiLevels L, const float *__restrict__ rois,
                                                                 const int *__restrict__ level, int C, int P, int sr,
                                                                 int aligned, float *__restrict__ out) {
  extern __shared__ float tile[];  // [nc][P*P]
  const int k = blockIdx.x, c0 = blockIdx.y * ROI_CCHUNK, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int PP = P * P, nc = min(ROI_CCHUNK, C - c0);
  // clamped: a level computed from a NaN / inf box area (diverged training) must not index past the table
  const int lv = level ? min(max(__builtin_amdgcn_readfirstlane(level[k]), 0), L.n - 1) : 0;
  const int H = L.H[lv], W = L.W[lv];
  const RoiGeom g = roi_geom(rois + (size_t)k * 5, L.scale[lv], P, sr, aligned);
  const float *base = L.feat[lv] + (size_t)g.b * H * W * C + c0;
  for (int bin = wave; bin < PP; bin += 4) {
    const int ph = bin / P, pw = bin - ph * P;
    for (int c = lane; c < nc; c += 64) {
      float acc = 0.f;
      for (int iy = 0; iy < g.gh; ++iy) {
        const float y = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
          const float x = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / g.gw;
          float v = 0.f;
          if (!(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W)) {
            float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
            int y0 = (int)yy, x0 = (int)xx, y1i, x1i;
            if (y0 >= H - 1) { y0 = y1i = H - 1; yy = (float)y0; } else y1i = y0 + 1;
            if (x0 >= W - 1) { x0 = x1i = W - 1; xx = (float)x0; } else x1i = x0 + 1;
            const float ly = yy - y0, lx = xx - x0, hy = 1.f - ly, hx = 1.f - lx;
            v = hy * hx * base[((size_t)y0 * W + x0) * C + c] + hy * lx * base[((size_t)y0 * W + x1i) * C + c] +
                ly * hx * base[((size_t)y1i * W + x0) * C + c] + ly * lx * base[((size_t)y1i * W + x1i) * C + c];
          }
          acc += v;
        }
      }
      tile[c * PP + bin] = acc / g.cnt;
    }
  }
  __syncthreads();
  float *dst = out + ((size_t)k * C + c0) * PP;
  for (int i = t; i < nc * PP; i += 256) dst[i] = tile[i];
}

// The same forward pass for a fixed sampling ratio (torchvision's MultiScaleRoIAlign uses 2) and C % 4 == 0: a lane owns FOUR
// consecutive channels (one 16-byte load per corner: a wave covers the 256 channels of a feature pixel in one instruction instead
// of four), the SR x SR x 4 corner loads of a bin are all issued before the first multiplication (one memory round trip per bin
// instead of SR x SR dependent ones), and the loops have constant trip counts.  Per element the same expression in the same
// order as above (bit-identical output).
template <int SR>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc_sr_kernel(RoiLevels L, const float *__restrict__ rois,
                                                                    const int *__restrict__ level, int C, int P, int aligned,
                                                                    float *__restrict__ out) {
  extern __shared__ float tile[];  // [nc][P*P]
  const int k = blockIdx.x, c0 = blockIdx.y * ROI_CCHUNK, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int PP = P * P, nc = min(ROI_CCHUNK, C - c0);
  const int lv = level ? min(max(__builtin_amdgcn_readfirstlane(level[k]), 0), L.n - 1) : 0;
  const int H = L.H[lv], W = L.W[lv];
  const RoiGeom g = roi_geom(rois + (size_t)k * 5, L.scale[lv], P, SR, aligned);
  const float *base = L.feat[lv] + (size_t)g.b * H * W * C + c0;
  const int c = lane * 4;
  if (c < nc) {
    for (int bin = wave; bin < PP; bin += 4) {
      const int ph = bin / P, pw = bin - ph * P;
      float4 v[SR * SR][4];
      float wy[SR][2], wx[SR][2];
      bool oky[SR], okx[SR];
      int ys[SR][2], xs[SR][2];
#pragma unroll
      for (int i = 0; i < SR; ++i) {
        const float y = g.y1 + ph * g.bh + (i + 0.5f) * g.bh / g.gh;
        oky[i] = !(y < -1.0f || y > (float)H);
        float yy = fmaxf(y, 0.f);
        int y0 = (int)yy, y1i;
        if (y0 >= H - 1) { y0 = y1i = H - 1; yy = (float)y0; } else y1i = y0 + 1;
        ys[i][0] = y0; ys[i][1] = y1i;
        wy[i][1] = yy - y0; wy[i][0] = 1.f - wy[i][1];
        const float x = g.x1 + pw * g.bw + (i + 0.5f) * g.bw / g.gw;
        okx[i] = !(x < -1.0f || x > (float)W);
        float xx = fmaxf(x, 0.f);
        int x0 = (int)xx, x1i;
        if (x0 >= W - 1) { x0 = x1i = W - 1; xx = (float)x0; } else x1i = x0 + 1;
        xs[i][0] = x0; xs[i][1] = x1i;
        wx[i][1] = xx - x0; wx[i][0] = 1.f - wx[i][1];
      }
#pragma unroll
      for (int iy = 0; iy < SR; ++iy)
#pragma unroll
        for (int ix = 0; ix < SR; ++ix)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            v[iy * SR + ix][q] = *reinterpret_cast<const float4 *>(base + ((size_t)ys[iy][q >> 1] * W + xs[ix][q & 1]) * C + c);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int iy = 0; iy < SR; ++iy)
#pragma unroll
        for (int ix = 0; ix < SR; ++ix) {
          const float4 *q = v[iy * SR + ix];
          const float hy = wy[iy][0], ly = wy[iy][1], hx = wx[ix][0], lx = wx[ix][1];
          const bool ok = oky[iy] && okx[ix];
          acc.x += ok ? hy * hx * q[0].x + hy * lx * q[1].x + ly * hx * q[2].x + ly * lx * q[3].x : 0.f;
          acc.y += ok ? hy * hx * q[0].y + hy * lx * q[1].y + ly * hx * q[2].y + ly * lx * q[3].y : 0.f;
          acc.z += ok ? hy * hx * q[0].z + hy * lx * q[1].z + ly * hx * q[2].z + ly * lx * q[3].z : 0.f;
          acc.w += ok ? hy * hx * q[0].w + hy * lx * q[1].w + ly * hx * q[2].w + ly * lx * q[3].w : 0.f;
        }
      tile[(c + 0) * PP + bin] = acc.x / g.cnt;
      tile[(c + 1) * PP + bin] = acc.y / g.cnt;
      tile[(c + 2) * PP + bin] = acc.z / g.cnt;
      tile[(c + 3) * PP + bin] = acc.w / g.cnt;
    }
  }
  __syncthreads();
  float *dst = out + ((size_t)k * C + c0) * PP;
  for (int i = t; i < nc * PP; i += 256) dst[i] = tile[i];
}

// Backward: one wave walks a ROW of samples (fixed y: two feature rows y0, y1 with weights hy, ly) from left to right and
// merges, in registers, what consecutive samples add to the same feature column -- at FPN's level assignment the sample
// spacing is 1-2 cells, so neighbouring samples share a column more often than not: a column's sum is flushed once, with two
// atomics (rows y0, y1), instead of four atomics per sample.  ~1.5x fewer L2 atomic operations (the kernel's bound: 822 M of
// them per train step at b = 8 before).  Per-element arithmetic: hy * sum(g * hx) instead of sum(g * hy * hx).
__global__ __launch_bounds__(256) void roi_align_bwd_nhwc_kernel(RoiLevels L, const float *__restrict__ rois,
                                                                 const int *__restrict__ level, int C, int P, int sr,
                                                                 int aligned, const float *__restrict__ gout) {
  extern __shared__ float tile[];
  const int k = blockIdx.x, c0 = blockIdx.y * ROI_CCHUNK, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int PP = P * P, nc = min(ROI_CCHUNK, C - c0);
  const float *src = gout + ((size_t)k * C + c0) * PP;
  for (int i = t; i < nc * PP; i += 256) tile[i] = src[i];
  __syncthreads();
  // clamped: a level computed from a NaN / inf box area (diverged training) must not index past the table
  const int lv = level ? min(max(__builtin_amdgcn_readfirstlane(level[k]), 0), L.n - 1) : 0;
  const int H = L.H[lv], W = L.W[lv];
  const RoiGeom g = roi_geom(rois + (size_t)k * 5, L.scale[lv], P, sr, aligned);
  float *base = L.gfeat[lv] + (size_t)g.b * H * W * C + c0;
  const float inv_cnt = 1.f / g.cnt;
  const int nrows = P * g.gh, ncols = P * g.gw;
  for (int row = wave; row < nrows; row += 4) {
    const int ph = row / g.gh, iy = row - ph * g.gh;
    const float y = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / g.gh;
    if (y < -1.0f || y > (float)H) continue;
    float yy = fmaxf(y, 0.f);
    int y0 = (int)yy, y1i;
    if (y0 >= H - 1) { y0 = y1i = H - 1; yy = (float)y0; } else y1i = y0 + 1;
    const float ly = yy - y0, hy = 1.f - ly;
    float *r0 = base + (size_t)y0 * W * C, *r1 = base + (size_t)y1i * W * C;
    for (int c = lane; c < nc; c += 64) {
      const float *gt = tile + c * PP + ph * P;
      int cur = -2;                 // column whose sum hA holds; hB: column cur + 1
      float hA = 0.f, hB = 0.f;
      auto flush = [&](int col, float h) {
        if (col < 0 || col >= W) return;
        atomicAdd(r0 + (size_t)col * C + c, hy * h);
        if (ly != 0.f) atomicAdd(r1 + (size_t)col * C + c, ly * h);
      };
      for (int px = 0; px < ncols; ++px) {
        const int pw = px / g.gw, ix = px - pw * g.gw;
        const float x = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / g.gw;
        if (x < -1.0f || x > (float)W) continue;
        float xx = fmaxf(x, 0.f);
        int x0 = (int)xx;
        bool edge = false;
        if (x0 >= W - 1) { x0 = W - 1; xx = (float)x0; edge = true; }
        const float lx = xx - x0, hx = 1.f - lx;
        if (x0 != cur) {            // wave-uniform: x does not depend on the lane
          if (cur >= 0) flush(cur, hA);
          if (x0 == cur + 1) hA = hB;
          else { if (cur >= 0) flush(cur + 1, hB); hA = 0.f; }
          hB = 0.f;
          cur = x0;
        }
        const float gv = gt[pw] * inv_cnt;
        hA += gv * hx;
        if (!edge) hB += gv * lx;
      }
      if (cur >= 0) { flush(cur, hA); flush(cur + 1, hB); }
    }
  }
}

// ---- COCO box IoU --------------------------------------------------------------------------------------
// pycocotools' bbIou (reference cocoapi/common/maskApi.c:109-120): boxes as (x, y, w, h) in float64,
// out[g * m + d] = intersection / union, with the union replaced by the DETECTION's area for crowd
// ground truth; 0 where the boxes do not overlap.  Same float64 expression order as the C code (and
// -ffp-contract=off), so the result is bit-identical.  One thread per (g, d) pair.
__global__ void coco_box_iou_kernel(const double *__restrict__ dt, const double *__restrict__ gt,
                                    const unsigned char *__restrict__ iscrowd, int m, int n, double *__restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)m * n) return;
  const int g = (int)(idx / m), d = (int)(idx - (long long)g * m);
  const double *G = gt + (size_t)g * 4, *D = dt + (size_t)d * 4;
  const double ga = G[2] * G[3], da = D[2] * D[3];
  const bool crowd = iscrowd != nullptr && iscrowd[g];
  double o = 0;
  const double w = fmin(D[2] + D[0], G[2] + G[0]) - fmax(D[0], G[0]);
  if (w > 0) {
    const double h = fmin(D[3] + D[1], G[3] + G[1]) - fmax(D[1], G[1]);
    if (h > 0) {
      const double i = w * h, u = crowd ? da : da + ga - i;
      o = i / u;
    }
  }
  out[idx] = o;
}

// ---- NMS ----------------------------------------------------------------------------------------
// Pass 1: 64 x 64 blocks of the upper-triangular suppression matrix as 64-bit masks (one wave per
// block, boxes of the column block staged in LDS).  Pass 2 (below) resolves them in score order.
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
  boxes += (size_t)blockIdx.z * n;               // one box set per blockIdx.z
  mask += (size_t)blockIdx.z * n * nblk;
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

// Pass 2.  One 1024-thread workgroup walks the boxes in score order, 64 at a time.  Per block of 64:
//   A. every wave resolves the 64 x 64 diagonal block on its own (lane l holds row l's diagonal word;
//      a uniform 64-step chain over `rem`, the removed-bits word of this block) -> kept bits `kb`;
//   B. the rows of the block were prefetched into registers one iteration earlier (they do not depend
//      on the outcome): thread (rg, w) ORs word w of its kept rows and merges it into the LDS
//      "removed" set with one ds_or_b64.
// WPR = words per row handled (power of two >= nblk), RG = 1024 / WPR row groups, RPT rows per thread.
template <int WPR>
__global__ __launch_bounds__(1024) void nms_reduce_kernel(const unsigned long long *__restrict__ mask, int n,
                                                          const unsigned char *__restrict__ valid,
                                                          long long *__restrict__ keep, int *__restrict__ count) {
  constexpr int RG = 1024 / WPR, RPT = 64 / RG;
  const int nblk = (n + 63) / 64, t = threadIdx.x, lane = t & 63;
  const int w = t % WPR, rg = t / WPR;
  mask += (size_t)blockIdx.x * n * nblk;          // one box set per workgroup
  keep += (size_t)blockIdx.x * n;
  count += blockIdx.x;
  __shared__ unsigned long long removed[256];
  if (t < 256) removed[t] = 0;
  if (valid) {  // boxes flagged invalid start out removed: never kept, never suppressing
    __syncthreads();
    valid += (size_t)blockIdx.x * n;
    for (int i = t; i < n; i += 1024)
      if (!valid[i]) atomicOr(&removed[i >> 6], 1ull << (i & 63));
  }
  unsigned long long cur[RPT], nxt[RPT], dcur, dnxt;
  auto fetch = [&](int blk, unsigned long long (&buf)[RPT], unsigned long long &diag) {
    const int row0 = blk * 64;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int row = row0 + rg + RG * k;
      buf[k] = (row < n && w > blk && w < nblk) ? mask[(size_t)row * nblk + w] : 0ull;
    }
    diag = (row0 + lane < n) ? mask[(size_t)(row0 + lane) * nblk + blk] : 0ull;
  };
  fetch(0, cur, dcur);
  int kept = 0;
  __syncthreads();
  for (int blk = 0; blk < nblk; ++blk) {
    if (blk + 1 < nblk) fetch(blk + 1, nxt, dnxt);
    unsigned long long rem = removed[blk];   // uniform: keep the chain below on the scalar unit
    rem = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(rem >> 32)) << 32) |
          (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)rem);   // the builtin returns int

    const int valid = min(64, n - blk * 64);
    if (valid < 64) rem |= ~0ull << valid;
    if (rem == ~0ull) {       // nothing of this block is left (padding / flagged invalid / suppressed): uniform, skip the chain
#pragma unroll
      for (int k = 0; k < RPT; ++k) cur[k] = nxt[k];
      dcur = dnxt;
      continue;
    }
    // row j's diagonal word only has bits > j, so bit j of `rem` is final when step j reads it
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((unsigned)dcur, j);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((unsigned)(dcur >> 32), j);
      if (!((rem >> j) & 1ull)) rem |= ((unsigned long long)hi << 32) | lo;
    }
    const unsigned long long kb = ~rem;
    unsigned long long acc = 0;
#pragma unroll
    for (int k = 0; k < RPT; ++k)
      if ((kb >> (rg + RG * k)) & 1ull) acc |= cur[k];
    if (acc) atomicOr(&removed[w], acc);
    if (t < 64 && ((kb >> t) & 1ull)) keep[kept + __popcll(kb & ((1ull << t) - 1ull))] = blk * 64 + t;
    kept += __popcll(kb);
#pragma unroll
    for (int k = 0; k < RPT; ++k) cur[k] = nxt[k];
    dcur = dnxt;
    __syncthreads();
  }
  if (t == 0) *count = kept;
  for (int i = kept + t; i < n; i += 1024) keep[i] = 0;  // defined tail: callers may gather through the padded list
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

static int fill_levels(RoiLevels &L, const void *const *ptrs, bool grads, const int *H, const int *W, const float *scale,
                       int n_levels, const char *who) {
  if (n_levels < 1 || n_levels > 4 || !ptrs || !H || !W || !scale) { set_error("%s: 1..4 levels with H, W, scale arrays", who); return DIB_EINVAL; }
  for (int i = 0; i < 4; ++i) {
    const int j = i < n_levels ? i : 0;
    if (!ptrs[j] || H[j] <= 0 || W[j] <= 0) { set_error("%s: level %d has a null pointer or empty shape", who, j); return DIB_EINVAL; }
    L.feat[i] = grads ? nullptr : (const float *)ptrs[j];
    L.gfeat[i] = grads ? (float *)const_cast<void *>(ptrs[j]) : nullptr;
    L.H[i] = H[j]; L.W[i] = W[j]; L.scale[i] = scale[j];
  }
  L.n = n_levels;
  return DIB_OK;
}

extern "C" int dib_roi_align_nhwc_forward(const float *const *feat_dev, const int *H, const int *W, const float *scale,
                                          int n_levels, const float *rois_dev, const int *level_dev, int K, int C, int pooled,
                                          int sampling_ratio, int aligned, float *out_dev, void *stream) {
  if (K < 0 || C <= 0 || pooled <= 0 || pooled > 7) { set_error("dib_roi_align_nhwc_forward: bad shape (pooled <= 7)"); return DIB_EINVAL; }
  if (K == 0) return DIB_OK;
  if (!rois_dev || !out_dev || (n_levels > 1 && !level_dev)) { set_error("dib_roi_align_nhwc_forward: null pointer"); return DIB_EINVAL; }
  RoiLevels L;
  int rc = fill_levels(L, (const void *const *)feat_dev, false, H, W, scale, n_levels, "dib_roi_align_nhwc_forward");
  if (rc != DIB_OK) return rc;
  const int nc = C < ROI_CCHUNK ? C : ROI_CCHUNK;
  bool vec = sampling_ratio == 2 && (C % 4) == 0;
  for (int i = 0; i < n_levels; ++i) vec = vec && (((uintptr_t)feat_dev[i]) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(roi_align_fwd_nhwc_sr_kernel<2>, dim3(K, (C + ROI_CCHUNK - 1) / ROI_CCHUNK), dim3(256),
                       (size_t)nc * pooled * pooled * sizeof(float), (hipStream_t)stream, L, rois_dev, level_dev, C, pooled, aligned, out_dev);
  else
    hipLaunchKernelGGL(roi_align_fwd_nhwc_kernel, dim3(K, (C + ROI_CCHUNK - 1) / ROI_CCHUNK), dim3(256),
                       (size_t)nc * pooled * pooled * sizeof(float), (hipStream_t)stream, L, rois_dev, level_dev, C, pooled,
                       sampling_ratio, aligned, out_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" int dib_roi_align_nhwc_backward(const float *grad_out_dev, const int *H, const int *W, const float *scale,
                                           int n_levels, const float *rois_dev, const int *level_dev, int K, int C, int pooled,
                                           int sampling_ratio, int aligned, float *const *grad_feat_dev, void *stream) {
  if (K < 0 || C <= 0 || pooled <= 0 || pooled > 7) { set_error("dib_roi_align_nhwc_backward: bad shape (pooled <= 7)"); return DIB_EINVAL; }
  if (K == 0) return DIB_OK;
  if (!rois_dev || !grad_out_dev || (n_levels > 1 && !level_dev)) { set_error("dib_roi_align_nhwc_backward: null pointer"); return DIB_EINVAL; }
  RoiLevels L;
  int rc = fill_levels(L, (const void *const *)grad_feat_dev, true, H, W, scale, n_levels, "dib_roi_align_nhwc_backward");
  if (rc != DIB_OK) return rc;
  const int nc = C < ROI_CCHUNK ? C : ROI_CCHUNK;
  hipLaunchKernelGGL(roi_align_bwd_nhwc_kernel, dim3(K, (C + ROI_CCHUNK - 1) / ROI_CCHUNK), dim3(256),
                     (size_t)nc * pooled * pooled * sizeof(float), (hipStream_t)stream, L, rois_dev, level_dev, C, pooled,
                     sampling_ratio, aligned, grad_out_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" int dib_coco_box_iou(const double *dt_dev, const double *gt_dev, const unsigned char *iscrowd_dev, int m, int n,
                                double *out_dev, void *stream) {
  if (m < 0 || n < 0) { set_error("dib_coco_box_iou: negative count"); return DIB_EINVAL; }
  if (m == 0 || n == 0) return DIB_OK;
  if (!dt_dev || !gt_dev || !out_dev) { set_error("dib_coco_box_iou: null pointer"); return DIB_EINVAL; }
  const long long total = (long long)m * n;
  hipLaunchKernelGGL(coco_box_iou_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dt_dev, gt_dev,
                     iscrowd_dev, m, n, out_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" size_t dib_nms_workspace_bytes(int n) {
  if (n <= 0) return 0;
  const size_t nblk = ((size_t)n + 63) / 64;
  return (size_t)n * nblk * sizeof(unsigned long long);
}

extern "C" int dib_nms_batched(const float *boxes_sorted_dev, const unsigned char *valid_dev, int B, int n,
                               float iou_threshold, void *workspace_dev, long long *keep_dev, int *count_dev, void *stream) {
  if (n < 0 || n > 16384 || B < 0) { set_error("dib_nms: n must be in [0, 16384] and B >= 0, got n=%d B=%d", n, B); return DIB_EINVAL; }
  if (B == 0) return DIB_OK;
  if (!count_dev) { set_error("dib_nms: null count pointer"); return DIB_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) { DIB_HIP_CHECK(hipMemsetAsync(count_dev, 0, sizeof(int) * (size_t)B, s)); return DIB_OK; }
  if (!boxes_sorted_dev || !workspace_dev || !keep_dev) { set_error("dib_nms: null pointer"); return DIB_EINVAL; }
  if (((uintptr_t)boxes_sorted_dev & 15) != 0) { set_error("dib_nms: boxes must be 16-byte aligned"); return DIB_EINVAL; }
  if (B > 65535) { set_error("dib_nms: at most 65535 box sets per call"); return DIB_EINVAL; }
  const int nblk = (n + 63) / 64;
  // no memset: pass 2 reads only the diagonal and upper blocks of rows < n, all written by pass 1
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nblk, nblk, B), dim3(64), 0, s, (const float4 *)boxes_sorted_dev, n, iou_threshold,
                     (unsigned long long *)workspace_dev);
  const unsigned long long *m = (const unsigned long long *)workspace_dev;
  if (nblk <= 16) hipLaunchKernelGGL(nms_reduce_kernel<16>, dim3(B), dim3(1024), 0, s, m, n, valid_dev, keep_dev, count_dev);
  else if (nblk <= 64) hipLaunchKernelGGL(nms_reduce_kernel<64>, dim3(B), dim3(1024), 0, s, m, n, valid_dev, keep_dev, count_dev);
  else hipLaunchKernelGGL(nms_reduce_kernel<256>, dim3(B), dim3(1024), 0, s, m, n, valid_dev, keep_dev, count_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" int dib_nms(const float *boxes_sorted_dev, int n, float iou_threshold, void *workspace_dev, long long *keep_dev,
                       int *count_dev, void *stream) {
  return dib_nms_batched(boxes_sorted_dev, nullptr, 1, n, iou_threshold, workspace_dev, keep_dev, count_dev, stream);
}
