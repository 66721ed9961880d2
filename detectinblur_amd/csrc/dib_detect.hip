// Box bookkeeping of the detector's training step as a handful of launches: IoU matching of candidates (anchors / proposals)
// against ragged ground truth, target encoding, proposal decoding.  In eager PyTorch these are ~60 (RPN) + ~45 (RoI heads) + ~25
// (decode) elementwise launches per step, in a stretch of the step where the HOST's launch rate is the limit
// (profiles/r4_train_step_conv.txt: ~10 us of wall per launch there): what torchvision's Matcher / BoxCoder / box_iou compute,
// restated with the same operations in the same order (no contraction: dib_common's build flags), so that thresholds fall the
// same way as in the tensor code they replace (detectinblur_amd/models/detector_ops.py keeps that code as the CPU path and the
// checker).  Reference: models/faster_rcnn.py:150-159,198-229 hands these parameters to torchvision's RPN / RoIHeads.
#include "dib_common.h"

namespace dib {

constexpr int MATCH_MAX_IMAGES = 32;
constexpr int MATCH_MAX_GT = 256;      // ground-truth boxes of one image held in LDS

struct GtOffsets { int off[MATCH_MAX_IMAGES + 1]; };

__device__ __forceinline__ float box_iou1(const float4 g, float area_g, const float4 c, float area_c) {
#pragma clang fp contract(off)
  // torchvision box_iou: lt = max(a[:2], b[:2]); rb = min(a[2:], b[2:]); wh = (rb - lt).clamp(min=0); inter / (area_a + area_b - inter)
  const float w = fmaxf(fminf(g.z, c.z) - fmaxf(g.x, c.x), 0.f), h = fmaxf(fminf(g.w, c.w) - fmaxf(g.y, c.y), 0.f);
  const float inter = w * h;
  return inter / (area_g + area_c - inter);
}
__device__ __forceinline__ float box_area1(const float4 b) {
#pragma clang fp contract(off)
  return (b.z - b.x) * (b.w - b.y);
}

// order-preserving map float -> unsigned (negative values below positive ones, NaN above everything: torch's max propagates NaN)
__device__ __forceinline__ unsigned ordered_key(float v) {
  const unsigned b = __float_as_uint(v);
  if (v != v) return 0xffffffffu;                      // either sign of NaN (0 / 0 comes out with the sign bit set on some paths)
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_value(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// pass 1 (only with allow_low_quality): best[t] = max over the image's candidates of IoU(gt t, candidate).  IoU >= 0 (or NaN for
// degenerate pairs, which the ordered key places above every number -- as torch's max propagates NaN)
__global__ __launch_bounds__(256) void box_best_per_gt_kernel(const float4 *__restrict__ gt, GtOffsets go, const float4 *__restrict__ cand, int M,
                                                             int shared, unsigned *__restrict__ best) {
  __shared__ float4 s_gt[MATCH_MAX_GT];
  __shared__ float s_area[MATCH_MAX_GT];
  __shared__ unsigned s_key[MATCH_MAX_GT];
  const int n = blockIdx.y, g0 = go.off[n], G = go.off[n + 1] - g0;
  if (G <= 0) return;
  for (int i = threadIdx.x; i < G; i += 256) { s_gt[i] = gt[g0 + i]; s_area[i] = box_area1(s_gt[i]); s_key[i] = 0u; }
  __syncthreads();
  const int m = blockIdx.x * 256 + threadIdx.x;
  const bool live = m < M;
  float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) c = cand[(shared ? 0 : (size_t)n * M) + m];
  const float ac = box_area1(c);
  for (int g = 0; g < G; ++g) {
    unsigned u = live ? ordered_key(box_iou1(s_gt[g], s_area[g], c, ac)) : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&s_key[g], u);
  }
  __syncthreads();
  // one global atomic per workgroup and ground truth, and only when it can raise the value: thousands of workgroups hammering the
  // same few addresses serialise in L2 (the first version: 0.5 ms for the RPN's 243k anchors)
  for (int i = threadIdx.x; i < G; i += 256) {
    const unsigned u = s_key[i];
    if (u > __atomic_load_n(best + g0 + i, __ATOMIC_RELAXED)) atomicMax(best + g0 + i, u);
  }
}

// pass 2: torchvision Matcher per candidate.  matches[n][m] = index of the ground truth of highest IoU (lowest index on ties), -1
// below `low`, -2 in [low, high); with allow_low every candidate that realises some ground truth's best IoU keeps its index.
// Images without ground truth: -1 everywhere.
__global__ __launch_bounds__(256) void box_match_kernel(const float4 *__restrict__ gt, GtOffsets go, const float4 *__restrict__ cand, int M, int shared,
                                                       float high, float low, int allow_low, const unsigned *__restrict__ best,
                                                       long long *__restrict__ match) {
  __shared__ float4 s_gt[MATCH_MAX_GT];
  __shared__ float s_area[MATCH_MAX_GT];
  __shared__ float s_best[MATCH_MAX_GT];
  const int n = blockIdx.y, g0 = go.off[n], G = go.off[n + 1] - g0;
  for (int i = threadIdx.x; i < G; i += 256) {
    s_gt[i] = gt[g0 + i];
    s_area[i] = box_area1(s_gt[i]);
    s_best[i] = allow_low ? ordered_value(best[g0 + i]) : 0.f;
  }
  __syncthreads();
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float4 c = cand[(shared ? 0 : (size_t)n * M) + m];
  const float ac = box_area1(c);
  float vmax = -1.f;
  int arg = 0;
  bool restore = false;
  for (int g = 0; g < G; ++g) {
    const float v = box_iou1(s_gt[g], s_area[g], c, ac);
    if (v > vmax || (v != v && vmax == vmax)) { vmax = v; arg = g; }      // NaN wins once, like torch.max
    restore = restore || (allow_low && v == s_best[g]);
  }
  long long r = arg;
  if (vmax < low) r = -1;
  else if (vmax >= low && vmax < high) r = -2;
  if (restore) r = arg;
  if (G <= 0) r = -1;
  match[(size_t)n * M + m] = r;
}

// BoxCoder.encode of the matched ground truth against its candidate (+ the matched box itself, optional):
//   dx = wx * (gx - px) / pw, dy = wy * (gy - py) / ph, dw = ww * log(gw / pw), dh = wh * log(gh / ph)
// match < 0 reads ground truth 0 (as `gt[m.clamp(min=0)]`); an image without ground truth reads a zero box.
__global__ __launch_bounds__(256) void box_encode_matched_kernel(const float4 *__restrict__ gt, GtOffsets go, const long long *__restrict__ match,
                                                                const float4 *__restrict__ cand, int M, int shared, float wx, float wy, float ww,
                                                                float wh, float4 *__restrict__ targets, float4 *__restrict__ matched) {
#pragma clang fp contract(off)
  const int n = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int g0 = go.off[n], G = go.off[n + 1] - g0;
  const long long mi = match[(size_t)n * M + m];
  const int gi = mi < 0 ? 0 : (int)mi;
  const float4 r = G > 0 ? gt[g0 + min(gi, G - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 p = cand[(shared ? 0 : (size_t)n * M) + m];
  const float pw = p.z - p.x, ph = p.w - p.y, px = p.x + 0.5f * pw, py = p.y + 0.5f * ph;
  const float gw = r.z - r.x, gh = r.w - r.y, gx = r.x + 0.5f * gw, gy = r.y + 0.5f * gh;
  if (targets) targets[(size_t)n * M + m] = make_float4(wx * (gx - px) / pw, wy * (gy - py) / ph, ww * logf(gw / pw), wh * logf(gh / ph));
  if (matched) matched[(size_t)n * M + m] = r;
}

// BoxCoder.decode for one box per row: deltas [R][4] against anchors [A][4] (row r uses anchor r % A)
__global__ __launch_bounds__(256) void box_decode_kernel(const float4 *__restrict__ deltas, const float4 *__restrict__ anchors, long long R, int A,
                                                        float wx, float wy, float ww, float wh, float clip, float4 *__restrict__ out) {
#pragma clang fp contract(off)
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float4 d = deltas[r], b = anchors[(int)(r % A)];
  const float w = b.z - b.x, h = b.w - b.y, cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
  // tensor / Python scalar is a multiplication by the float reciprocal in ATen (BinaryDivTrueKernel: `a * (1 / b)`), not a division
  float dw = d.z * (1.f / ww), dh = d.w * (1.f / wh);
  dw = dw > clip ? clip : dw; dh = dh > clip ? clip : dh;          // torch.clamp(max=clip): NaN stays NaN
  const float dx = d.x * (1.f / wx), dy = d.y * (1.f / wy);
  const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
  out[r] = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
}

// RoI-head candidate pool: cands[n] = proposals[n] (P rows) ++ ground truth of image n ++ [0, 0, 1, 1] padding up to P + Gpad rows
__global__ __launch_bounds__(256) void box_pool_kernel(const float4 *__restrict__ props, int P, const float4 *__restrict__ gt, GtOffsets go, int Gpad,
                                                      float4 *__restrict__ cands) {
  const int n = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x, M = P + Gpad;
  if (j >= M) return;
  const int g0 = go.off[n], G = go.off[n + 1] - g0;
  float4 v = make_float4(0.f, 0.f, 1.f, 1.f);
  if (j < P) v = props[(size_t)n * P + j];
  else if (j - P < G) v = gt[g0 + j - P];
  cands[(size_t)n * M + j] = v;
}

// class of every pool row: gt_labels[match] for matches >= 0, 0 for BELOW_LOW, -1 for BETWEEN and for rows that are padding
// (proposal rows with ok == 0, ground-truth rows beyond the image's own count)
__global__ __launch_bounds__(256) void box_labels_kernel(const long long *__restrict__ match, const long long *__restrict__ gt_labels, GtOffsets go,
                                                        const unsigned char *__restrict__ ok, int P, int M, long long *__restrict__ labels) {
  const int n = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  const int g0 = go.off[n], G = go.off[n + 1] - g0;
  const bool live = j < P ? (ok ? ok[(size_t)n * P + j] != 0 : true) : (j - P < G);
  const long long m = match[(size_t)n * M + j];
  long long lab = m >= 0 ? (G > 0 ? gt_labels[g0 + min((int)m, G - 1)] : 0) : (m == -1 ? 0 : -1);
  labels[(size_t)n * M + j] = live ? lab : -1;
}

}  // namespace dib

using namespace dib;

static int check_match_args(const char *who, const float *gt, const int *off, int N, const float *cand, int M) {
  if (N <= 0 || N > MATCH_MAX_IMAGES || M < 0 || !off || !cand) { set_error("%s: 1..%d images, candidates and offsets required", who, MATCH_MAX_IMAGES); return DIB_EINVAL; }
  for (int n = 0; n < N; ++n)
    if (off[n + 1] < off[n] || off[n + 1] - off[n] > MATCH_MAX_GT) { set_error("%s: image %d has %d ground-truth boxes (0..%d supported)", who, n, off[n + 1] - off[n], MATCH_MAX_GT); return DIB_EINVAL; }
  if (off[N] > off[0] && !gt) { set_error("%s: null ground truth", who); return DIB_EINVAL; }
  if ((((uintptr_t)gt | (uintptr_t)cand) & 15) != 0) { set_error("%s: boxes must be 16-byte aligned", who); return DIB_EINVAL; }
  return DIB_OK;
}

extern "C" int dib_box_match(const float *gt_cat_dev, const int *gt_offset, int N, const float *cand_dev, int M, int cand_shared, float high,
                             float low, int allow_low_quality, unsigned *best_dev, long long *match_dev, void *stream) {
  if (int rc = check_match_args("dib_box_match", gt_cat_dev, gt_offset, N, cand_dev, M)) return rc;
  if (M == 0) return DIB_OK;
  if (!match_dev || (allow_low_quality && !best_dev)) { set_error("dib_box_match: null output / workspace"); return DIB_EINVAL; }
  GtOffsets go;
  for (int n = 0; n <= N; ++n) go.off[n] = gt_offset[n];
  const dim3 grid((M + 255) / 256, N);
  hipStream_t s = (hipStream_t)stream;
  if (allow_low_quality && go.off[N] > go.off[0]) {
    DIB_HIP_CHECK(hipMemsetAsync(best_dev + go.off[0], 0, (size_t)(go.off[N] - go.off[0]) * sizeof(unsigned), s));
    hipLaunchKernelGGL(box_best_per_gt_kernel, grid, dim3(256), 0, s, (const float4 *)gt_cat_dev, go, (const float4 *)cand_dev, M, cand_shared, best_dev);
  }
  hipLaunchKernelGGL(box_match_kernel, grid, dim3(256), 0, s, (const float4 *)gt_cat_dev, go, (const float4 *)cand_dev, M, cand_shared, high, low,
                     allow_low_quality, best_dev, match_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" int dib_box_encode_matched(const float *gt_cat_dev, const int *gt_offset, int N, const long long *match_dev, const float *cand_dev, int M,
                                      int cand_shared, float wx, float wy, float ww, float wh, float *targets_dev, float *matched_dev, void *stream) {
  if (int rc = check_match_args("dib_box_encode_matched", gt_cat_dev, gt_offset, N, cand_dev, M)) return rc;
  if (M == 0) return DIB_OK;
  if (!match_dev || (!targets_dev && !matched_dev)) { set_error("dib_box_encode_matched: null pointer"); return DIB_EINVAL; }
  if ((((uintptr_t)targets_dev | (uintptr_t)matched_dev) & 15) != 0) { set_error("dib_box_encode_matched: outputs must be 16-byte aligned"); return DIB_EINVAL; }
  GtOffsets go;
  for (int n = 0; n <= N; ++n) go.off[n] = gt_offset[n];
  hipLaunchKernelGGL(box_encode_matched_kernel, dim3((M + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, (const float4 *)gt_cat_dev, go, match_dev,
                     (const float4 *)cand_dev, M, cand_shared, wx, wy, ww, wh, (float4 *)targets_dev, (float4 *)matched_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" int dib_box_decode(const float *deltas_dev, const float *anchors_dev, long long R, int A, float wx, float wy, float ww, float wh,
                              float clip, float *out_dev, void *stream) {
  if (R < 0 || A <= 0) { set_error("dib_box_decode: bad sizes"); return DIB_EINVAL; }
  if (R == 0) return DIB_OK;
  if (!deltas_dev || !anchors_dev || !out_dev || ((((uintptr_t)deltas_dev | (uintptr_t)anchors_dev | (uintptr_t)out_dev)) & 15) != 0) {
    set_error("dib_box_decode: null or misaligned pointer");
    return DIB_EINVAL;
  }
  hipLaunchKernelGGL(box_decode_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)deltas_dev,
                     (const float4 *)anchors_dev, R, A, wx, wy, ww, wh, clip, (float4 *)out_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" int dib_box_pool(const float *proposals_dev, int P, const float *gt_cat_dev, const int *gt_offset, int N, int Gpad, float *cands_dev,
                            void *stream) {
  if (int rc = check_match_args("dib_box_pool", gt_cat_dev, gt_offset, N, cands_dev, P + Gpad)) return rc;
  if (P < 0 || Gpad < 0 || (P > 0 && !proposals_dev) || (((uintptr_t)proposals_dev) & 15) != 0) { set_error("dib_box_pool: bad proposals"); return DIB_EINVAL; }
  for (int n = 0; n < N; ++n)
    if (gt_offset[n + 1] - gt_offset[n] > Gpad) { set_error("dib_box_pool: image %d has more ground truth than the pool's %d rows", n, Gpad); return DIB_EINVAL; }
  if (P + Gpad == 0) return DIB_OK;
  GtOffsets go;
  for (int n = 0; n <= N; ++n) go.off[n] = gt_offset[n];
  hipLaunchKernelGGL(box_pool_kernel, dim3((P + Gpad + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, (const float4 *)proposals_dev, P,
                     (const float4 *)gt_cat_dev, go, Gpad, (float4 *)cands_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" int dib_box_labels(const long long *match_dev, const long long *gt_labels_cat_dev, const int *gt_offset, int N, const unsigned char *ok_dev,
                              int P, int M, long long *labels_dev, void *stream) {
  if (N <= 0 || N > MATCH_MAX_IMAGES || !gt_offset || P < 0 || M < P || !match_dev || !labels_dev) { set_error("dib_box_labels: bad arguments"); return DIB_EINVAL; }
  if (gt_offset[N] > gt_offset[0] && !gt_labels_cat_dev) { set_error("dib_box_labels: null labels"); return DIB_EINVAL; }
  if (M == 0) return DIB_OK;
  GtOffsets go;
  for (int n = 0; n <= N; ++n) go.off[n] = gt_offset[n];
  hipLaunchKernelGGL(box_labels_kernel, dim3((M + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, match_dev, gt_labels_cat_dev, go, ok_dev, P, M,
                     labels_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

// ---- detections of one image: softmax + per-class box decoding + clipping + score / size tests in one launch --------------------
// (torchvision RoIHeads.postprocess_detections up to the NMS; reference models/faster_rcnn.py:213-229 sets score_thresh 0.05,
// nms_thresh 0.5, detections_per_img 100 and the (10, 10, 5, 5) weights).  One wave per RoI, lane l owns classes l and l + 64.
// The softmax repeats ATen's softmax_warp_forward for <= 128 classes operation by operation (maximum, exp(x - max), the two
// addends of a lane first, then the xor-butterfly from offset 32 down to 1, exp(..) / sum), so scores land on the same side of the
// threshold as the eager path's.  Output is class-major: row c - 1 holds class c's score (-inf where the candidate is dropped) and
// clipped box for every RoI.
namespace dib {

__global__ __launch_bounds__(256) void det_candidates_kernel(const float *__restrict__ logits, const float4 *__restrict__ deltas,
                                                            const float4 *__restrict__ rois, int R, int C, float img_h, float img_w, float wx,
                                                            float wy, float ww, float wh, float clip, float score_thresh, float min_size,
                                                            float *__restrict__ scores_cm, float4 *__restrict__ boxes_cm,
                                                            unsigned *__restrict__ stats) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int c0 = lane, c1 = lane + 64;
  const float x0 = c0 < C ? logits[(size_t)r * C + c0] : -INFINITY, x1 = c1 < C ? logits[(size_t)r * C + c1] : -INFINITY;
  float m = x0 < x1 ? x1 : x0;                       // (a < b) ? b : a, ATen's Max functor
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const float b = __shfl_xor(m, o, 64); m = m < b ? b : m; }
  const float e0 = expf(x0 - m), e1 = expf(x1 - m);
  float s = 0.f;
  s += e0;
  s += e1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s = s + __shfl_xor(s, o, 64);
  const float4 p = rois[r];
  const float w = p.z - p.x, h = p.w - p.y, cx = p.x + 0.5f * w, cy = p.y + 0.5f * h;
  int n_valid = 0;
  float big = 0.f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int c = half ? c1 : c0;
    if (c >= 1 && c < C) {
      const float score = (half ? e1 : e0) / s;
      const float4 d = deltas[(size_t)r * C + c];
      float dw = d.z * (1.f / ww), dh = d.w * (1.f / wh);
      dw = dw > clip ? clip : dw; dh = dh > clip ? clip : dh;
      const float dx = d.x * (1.f / wx), dy = d.y * (1.f / wy);
      const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
      float4 b = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
      b.x = b.x < 0.f ? 0.f : b.x; b.x = b.x > img_w ? img_w : b.x;        // clamp(min=0, max=w): NaN stays NaN
      b.z = b.z < 0.f ? 0.f : b.z; b.z = b.z > img_w ? img_w : b.z;
      b.y = b.y < 0.f ? 0.f : b.y; b.y = b.y > img_h ? img_h : b.y;
      b.w = b.w < 0.f ? 0.f : b.w; b.w = b.w > img_h ? img_h : b.w;
      const bool ok = score > score_thresh && (b.z - b.x) >= min_size && (b.w - b.y) >= min_size;
      scores_cm[(size_t)(c - 1) * R + r] = ok ? score : -INFINITY;
      boxes_cm[(size_t)(c - 1) * R + r] = b;
      if (ok) { ++n_valid; big = fmaxf(big, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))); }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { n_valid += __shfl_xor(n_valid, o, 64); big = fmaxf(big, __shfl_xor(big, o, 64)); }
  if (lane == 0 && n_valid) {
    atomicAdd(stats, (unsigned)n_valid);
    atomicMax(stats + 1, __float_as_uint(big));       // coordinates are >= 0 after clipping: the bit patterns order like the numbers
  }
}

}  // namespace dib

extern "C" int dib_det_candidates(const float *logits_dev, const float *deltas_dev, const float *rois_dev, int R, int C, float img_h, float img_w,
                                  float wx, float wy, float ww, float wh, float clip, float score_thresh, float min_size, float *scores_cm_dev,
                                  float *boxes_cm_dev, unsigned *stats_dev, void *stream) {
  if (R < 0 || C < 2 || C > 128) { set_error("dib_det_candidates: 2..128 classes"); return DIB_EINVAL; }
  if (!stats_dev) { set_error("dib_det_candidates: null stats"); return DIB_EINVAL; }
  DIB_HIP_CHECK(hipMemsetAsync(stats_dev, 0, 2 * sizeof(unsigned), (hipStream_t)stream));
  if (R == 0) return DIB_OK;
  if (!logits_dev || !deltas_dev || !rois_dev || !scores_cm_dev || !boxes_cm_dev ||
      ((((uintptr_t)deltas_dev | (uintptr_t)rois_dev | (uintptr_t)boxes_cm_dev)) & 15) != 0) {
    set_error("dib_det_candidates: null or misaligned pointer");
    return DIB_EINVAL;
  }
  hipLaunchKernelGGL(dib::det_candidates_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits_dev, (const float4 *)deltas_dev,
                     (const float4 *)rois_dev, R, C, img_h, img_w, wx, wy, ww, wh, clip, score_thresh, min_size, scores_cm_dev,
                     (float4 *)boxes_cm_dev, stats_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
