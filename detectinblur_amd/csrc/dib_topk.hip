// Sorted top-k of every (image, pyramid level) row of RPN scores in ONE launch, with the winners' boxes gathered, clipped to the image
// and flagged (too small = takes no part in the NMS) by the same kernel.  What it replaces in the proposal filter
// (torchvision RegionProposalNetwork.filter_proposals as models/rpn.py restates it; reference models/faster_rcnn.py:198-207 sets
// pre_nms_top_n / post_nms_top_n): per level a torch.topk (13-15 launches of ATen's multi-block radix select for the big levels),
// two slice copies and a gather, then ~10 elementwise launches of clipping and size tests -- ~115 launches per call, twice per
// training step / once per evaluated image, in the stretch of the step where launches, not bytes, are the cost
// (profiles/r4_train_step_conv.txt, profiles/r4_trunk_b1_trace.txt).
//
// One workgroup of 1024 threads per row.  Radix select on an order-preserving 32-bit key (11 + 11 + 10 bits, histogram in LDS with
// the lanes of a wave that hit the same bin adding once), one more sweep that collects the k winners into LDS, bitonic sort of
// (key, index) pairs there.  A row of 201,600 scores (P2 at 800 x 1333) takes ~170 us, bound by the instruction rate of the ONE compute
// unit that sweeps it four times (~20 instructions per element; a lower bound from per-thread maxima that spares most elements the
// histogram atomics was tried and cost more than it saved: the sweeps are not atomics-bound) -- against ~350 us and 75 launches for
// ATen's chain; callers split a long row into consecutive pieces (each a level of its own: up to 32 per launch) and merge the pieces' winners with a
// second launch (models/detector_ops.py: topk_levels_split_hip) -- the 201,600-score row in 13 pieces: 66 us for both launches.  Order: descending score, ascending index among equal scores (torch.topk leaves the order of ties
// unspecified; this one is deterministic), NaN above every number (as torch.topk ranks it).
#include "dib_common.h"

namespace dib {

constexpr int TOPK_MAX_LEVELS = 32;
constexpr int TOPK_MAX_K = 2048;
constexpr int TOPK_THREADS = 1024;

struct TopkLevels {
  int off[TOPK_MAX_LEVELS + 1];   // row-relative start of every level (elements)
  int k[TOPK_MAX_LEVELS];         // winners wanted per level (<= K)
};

__device__ __forceinline__ unsigned topk_key(float v) {
  const unsigned b = __float_as_uint(v);
  if (v != v) return 0xffffffffu;
  if (v == 0.f) return 0x80000000u;                       // -0.0 ranks with +0.0, as it compares
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// histogram add.  Scores of a level crowd into a few of the top-bit bins, the worst case for LDS atomics (64 same-address adds of
// one wave instruction serialise): a wave whose active lanes all hit one bin adds once; otherwise every lane adds for itself (a loop
// over the distinct bins of the wave cost 10 x more than the conflicts it avoided when the bins were diverse: measured)
__device__ __forceinline__ void hist_add(unsigned *hist, bool active, unsigned bin) {
  if (active) {                                    // scalar broadcast of the first active lane's bin: no trip through the LDS crossbar
    const unsigned b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)bin);
    const unsigned long long m = __ballot(true), same = __ballot(bin == b0);
    if (same == m) {
      if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&hist[b0], (unsigned)__popcll(m));
    } else atomicAdd(&hist[bin], 1u);
  }
}

constexpr int TOPK_UNROLL = 8;      // loads in flight per thread in the sweeps over a row (one per iteration: 0.7 us of latency each)

__global__ __launch_bounds__(TOPK_THREADS) void topk_levels_kernel(const float *__restrict__ values, long long row_stride, TopkLevels lv, int K,
                                                                  const float4 *__restrict__ boxes, const float *__restrict__ clip_wh,
                                                                  float min_size, float *__restrict__ out_scores,
                                                                  long long *__restrict__ out_index, float4 *__restrict__ out_boxes,
                                                                  unsigned char *__restrict__ out_valid) {
#pragma clang fp contract(off)
  __shared__ unsigned hist[2048];
  __shared__ unsigned long long buf[TOPK_MAX_K];
  __shared__ unsigned s_wave[TOPK_THREADS / 64];
  __shared__ unsigned s_bin, s_above, s_count, s_eq_taken;
  const int l = blockIdx.x, L = gridDim.x, n = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cnt = lv.off[l + 1] - lv.off[l];
  const int k = min(min(lv.k[l], cnt), K);
  const float *row = values + (size_t)n * row_stride + lv.off[l];
  const size_t out0 = ((size_t)n * L + l) * K;

  unsigned thr = 0;           // key of the k-th largest element
  unsigned need_eq = 0;       // how many elements equal to it are among the winners
  unsigned count_eq = 0;      // how many there are
  if (k > 0 && k < cnt) {
    unsigned prefix = 0, mask = 0, need = (unsigned)k;
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0), bits = pass == 2 ? 10 : 11;
      for (int i = t; i < 2048; i += TOPK_THREADS) hist[i] = 0;
      __syncthreads();
      for (int i0 = 0; i0 < cnt; i0 += TOPK_THREADS * TOPK_UNROLL) {
        float v[TOPK_UNROLL];
#pragma unroll
        for (int u = 0; u < TOPK_UNROLL; ++u) { const int i = i0 + u * TOPK_THREADS + t; v[u] = i < cnt ? row[i] : 0.f; }
#pragma unroll
        for (int u = 0; u < TOPK_UNROLL; ++u) {
          const unsigned key = topk_key(v[u]);
          const bool act = i0 + u * TOPK_THREADS + t < cnt && (key & mask) == prefix;
          hist_add(hist, act, (key >> shift) & ((1u << bits) - 1u));
        }
      }
      __syncthreads();
      // the bin b with (elements in bins above b) < need <= (elements in bins >= b): suffix sums, two bins per thread
      const unsigned h1 = hist[2047 - 2 * t], h0 = hist[2046 - 2 * t];       // descending bin order: thread t owns bins 2047-2t, 2046-2t
      unsigned s = h1 + h0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned v = (unsigned)__shfl_up((int)s, o, 64); if (lane >= o) s += v; }
      if (lane == 63) s_wave[wave] = s;
      __syncthreads();
      unsigned before = 0;
      for (int w = 0; w < wave; ++w) before += s_wave[w];
      const unsigned incl = before + s, excl = incl - (h1 + h0);            // elements in bins above this thread's pair / incl. it
      if (excl < need && need <= incl) {
        const bool first = need <= excl + h1;
        s_bin = first ? 2047 - 2 * t : 2046 - 2 * t;
        s_above = first ? excl : excl + h1;
      }
      __syncthreads();
      const unsigned b = s_bin;
      need -= s_above;
      prefix |= b << shift;
      mask |= ((1u << bits) - 1u) << shift;
      if (pass == 2) count_eq = hist[b];
      __syncthreads();
    }
    thr = prefix;
    need_eq = need;
  }

  // collect the winners: everything above the threshold, and of its equals the `need_eq` lowest indices (all of them in the common
  // case of no tie at the threshold)
  if (t == 0) { s_count = 0; s_eq_taken = 0; }
  __syncthreads();
  const bool take_all = !(k > 0 && k < cnt);
  const bool ordered_eq = !take_all && count_eq != need_eq;
  if (k > 0) {
    for (int i00 = 0; i00 < cnt; i00 += TOPK_THREADS * TOPK_UNROLL) {
     float v[TOPK_UNROLL];
#pragma unroll
     for (int u = 0; u < TOPK_UNROLL; ++u) { const int i = i00 + u * TOPK_THREADS + t; v[u] = i < cnt ? row[i] : 0.f; }
#pragma unroll
     for (int u = 0; u < TOPK_UNROLL; ++u) {
      const int i0 = i00 + u * TOPK_THREADS;
      if (i0 >= cnt) break;                       // uniform
      const int i = i0 + t;
      unsigned key = 0;
      bool in = i < cnt, sel = false, eq = false;
      if (in) {
        key = topk_key(v[u]);
        sel = take_all || key > thr;
        eq = !take_all && key == thr;
      }
      if (!ordered_eq) sel = sel || eq;
      else {
        // rare: more equals than places -- rank them by index across the workgroup
        const unsigned long long m = __ballot(eq);
        if (lane == 0) s_wave[wave] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned rank = s_eq_taken + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        unsigned total = 0;
        for (int w = 0; w < TOPK_THREADS / 64; ++w) { if (w < wave) rank += s_wave[w]; total += s_wave[w]; }
        if (eq && rank < need_eq) sel = true;
        __syncthreads();
        if (t == 0) s_eq_taken += total;
      }
      if (sel) {
        const unsigned long long m = __ballot(true);
        unsigned base = 0;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&s_count, (unsigned)__popcll(m));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < TOPK_MAX_K) buf[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
      }
      if (ordered_eq) __syncthreads();
     }
    }
  }
  __syncthreads();
  int P = 1;
  while (P < k) P <<= 1;
  for (int i = k + t; i < P; i += TOPK_THREADS) buf[i] = 0ull;       // below every real pair (keys are >= 0x007fffff)
  // bitonic sort, descending
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int p = t; p < (P >> 1); p += TOPK_THREADS) {
        const int i = 2 * p - (p & (stride - 1)), j = i + stride;
        const bool desc = (i & size) == 0;
        const unsigned long long a = buf[i], b = buf[j];
        if ((a < b) == desc) { buf[i] = b; buf[j] = a; }
      }
    }
  }
  __syncthreads();
  float cw = 0.f, ch = 0.f;
  if (clip_wh) { cw = clip_wh[2 * n]; ch = clip_wh[2 * n + 1]; }
  for (int j = t; j < K; j += TOPK_THREADS) {
    float sc = -INFINITY;
    long long idx = 0;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned char ok = 0;
    if (j < k) {
      idx = (long long)(0xffffffffu - (unsigned)buf[j]);
      sc = row[idx];
      if (boxes) {
        bx = boxes[(size_t)n * row_stride + lv.off[l] + idx];
        if (clip_wh) {       // clamp(min=0).minimum(size): a NaN coordinate stays NaN
          bx.x = bx.x < 0.f ? 0.f : bx.x; bx.x = bx.x > cw ? cw : bx.x;
          bx.z = bx.z < 0.f ? 0.f : bx.z; bx.z = bx.z > cw ? cw : bx.z;
          bx.y = bx.y < 0.f ? 0.f : bx.y; bx.y = bx.y > ch ? ch : bx.y;
          bx.w = bx.w < 0.f ? 0.f : bx.w; bx.w = bx.w > ch ? ch : bx.w;
        }
        ok = sc > -INFINITY && (bx.z - bx.x) >= min_size && (bx.w - bx.y) >= min_size;
      }
    }
    out_scores[out0 + j] = sc;
    if (out_index) out_index[out0 + j] = idx;
    if (out_boxes) out_boxes[out0 + j] = bx;
    if (out_valid) out_valid[out0 + j] = ok;
  }
}

}  // namespace dib

using namespace dib;

extern "C" int dib_topk_levels(const float *values_dev, long long row_stride, int N, const int *level_offset, const int *level_k, int L, int K,
                               const float *boxes_dev, const float *clip_wh_dev, float min_size, float *out_scores_dev, long long *out_index_dev,
                               float *out_boxes_dev, unsigned char *out_valid_dev, void *stream) {
  if (N < 0 || L <= 0 || L > TOPK_MAX_LEVELS || K <= 0 || K > TOPK_MAX_K || !level_offset || !level_k) {
    set_error("dib_topk_levels: 1..%d levels, 1..%d winners per level", TOPK_MAX_LEVELS, TOPK_MAX_K);
    return DIB_EINVAL;
  }
  TopkLevels lv;
  for (int l = 0; l <= L; ++l) lv.off[l] = level_offset[l];
  for (int l = 0; l < L; ++l) {
    if (level_offset[l + 1] < level_offset[l] || level_k[l] < 0 || level_k[l] > K) { set_error("dib_topk_levels: bad level %d", l); return DIB_EINVAL; }
    lv.k[l] = level_k[l];
  }
  if (level_offset[0] < 0 || level_offset[L] > row_stride) { set_error("dib_topk_levels: the levels [%d, %d) do not lie inside a row of %lld", level_offset[0], level_offset[L], row_stride); return DIB_EINVAL; }
  if (N == 0) return DIB_OK;
  if (!values_dev || !out_scores_dev || ((out_boxes_dev || out_valid_dev) && !boxes_dev)) { set_error("dib_topk_levels: null pointer"); return DIB_EINVAL; }
  if ((((uintptr_t)boxes_dev | (uintptr_t)out_boxes_dev) & 15) != 0) { set_error("dib_topk_levels: boxes must be 16-byte aligned"); return DIB_EINVAL; }
  hipLaunchKernelGGL(topk_levels_kernel, dim3(L, N), dim3(TOPK_THREADS), 0, (hipStream_t)stream, values_dev, row_stride, lv, K, (const float4 *)boxes_dev,
                     clip_wh_dev, min_size, out_scores_dev, out_index_dev, (float4 *)out_boxes_dev, out_valid_dev);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
