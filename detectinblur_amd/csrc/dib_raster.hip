// PSF rasteriser for gfx950: trajectory -> float64 PSF -> (centre) -> (crop) -> float64 / float16.
//
// Reference: motion_blur/generate_PSF.py:31-83 (`PSF.fit`: a 2000-iteration Python loop, ~90 ms per
// PSF on one CPU core) and :106-123 (`centerPSF`), then the [64:192] crop of transforms.py:334-335
// and the float64 -> Half conversion of engine.py:84.
//
// Bit-exactness: float64 addition is not associative, so the reference's order is kept where it
// matters.  (1) Each PSF cell receives its bilinear splats in ascending sample order: one workgroup
// owns one canvas row, lane = column; the samples that touch the row are found 256 at a time with a
// ballot and visited in bit order.  A lane evaluates  w_t * (tri(re - col) * tri(im - row))  -- the
// reference's own expression for whichever of the four splats lands on its cell -- so no scatter and
// no atomics are needed.  Samples with zero exposure weight are skipped (adding +0.0 is exact).
// (2) `np.sum` is reproduced structurally: 8192-element iterator chunks added left to right, each
// chunk pairwise-summed down to 128-element leaves of 8 interleaved accumulators (numpy's
// DOUBLE_pairwise_sum).  (3) The centroid is accumulated sequentially in np.nonzero (row-major)
// order by one lane, from products computed in parallel.
#include "dib_common.h"
#include <hip/hip_fp16.h>

namespace dib {

__device__ inline double tri(double u) { return fmax(0.0, 1.0 - fabs(u)); }

// exposure weight of sample t for the window (0, frac]  (generate_PSF.py:47-56 with prevT = 0)
__device__ inline double sample_weight(int t, double fn) {
  if (t <= 0) return 0.0;
  if (t == 1) return fn >= 1.0 ? 1.0 : (fn >= 0.0 ? fn : 0.0);
  if (fn >= (double)t) return 1.0;
  if (fn >= (double)(t - 1)) return fn - (double)(t - 1);
  return 0.0;
}

// fraction * iters per PSF travels in the kernel-argument buffer (no H2D copy, capture-safe)
constexpr int RASTER_CHUNK = 64;
struct FnBatch { double fn[RASTER_CHUNK]; };

// grid (canvas rows, PSFs of this chunk); block 256 threads; thread = column (canvas <= 256)
__global__ __launch_bounds__(256) void raster_rows_kernel(const double2 *__restrict__ traj, int iters, FnBatch fnb,
                                                          int canvas, double *__restrict__ raw) {
#pragma clang fp contract(off)
  const int row = blockIdx.x, b = blockIdx.y, col = threadIdx.x;
  const double2 *x = traj + (size_t)b * iters;
  const double fn = fnb.fn[b];
  // samples with t > fn + 1 have zero weight
  int nsamp = iters;
  if (fn + 2.0 < (double)iters) nsamp = (int)fn + 2;
  if (nsamp < 0) nsamp = 0;
  __shared__ unsigned long long s_mask[4];
  double acc = 0.0;
  for (int base = 0; base < nsamp; base += 256) {
    const int t = base + threadIdx.x;
    bool hit = false;
    if (t < nsamp) {
      const double im = x[t].y;
      int m1 = (int)floor(im);
      m1 = min(canvas - 1, max(1, m1));
      hit = (m1 == row) || (m1 + 1 == row);
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0) s_mask[threadIdx.x >> 6] = m;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      unsigned long long mm = s_mask[w];
      while (mm) {
        const int bit = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        const int tt = base + w * 64 + bit;
        const double2 p = x[tt];
        int m2 = (int)floor(p.x);
        m2 = min(canvas - 1, max(1, m2));
        const int dx = col - m2;
        if (dx == 0 || dx == 1) {
          const double wt = sample_weight(tt, fn);
          const double prod = tri(p.x - (double)col) * tri(p.y - (double)row);
          acc += wt * prod;
        }
      }
    }
    __syncthreads();
  }
  if (col < canvas) raw[((size_t)b * canvas + row) * canvas + col] = acc / (double)iters;  // :77
}

// one block per PSF: np.sum, centroid, offsets; then roll + crop + convert
__global__ __launch_bounds__(256) void raster_finish_kernel(const double *__restrict__ raw_all, int canvas, int center,
                                                            int out_n, double *__restrict__ out64,
                                                            __half *__restrict__ out16, double2 *__restrict__ work_all) {
#pragma clang fp contract(off)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = canvas * canvas;
  const double *raw = raw_all + (size_t)b * n;
  double2 *work = work_all + (size_t)b * n;
  __shared__ double s_blk[512];
  __shared__ double s_total;
  __shared__ int s_cnt[1024 + 1];
  __shared__ int s_off[2];
  __shared__ double2 s_stage[1024];

  if (center) {
    // ---- np.sum(raw): 128-element leaves -> pairwise tree per 8192-chunk -> sequential chunks ----
    const int nleaf = n / 128;  // canvas in {64,128,256}: 32, 128, 512
    for (int l = tid; l < nleaf; l += 256) {
      const double *a = raw + (size_t)l * 128;
      double r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = a[k];
      for (int i = 8; i < 128; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] += a[i + k];
      }
      s_blk[l] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    }
    __syncthreads();
    const int leaves_per_chunk = min(nleaf, 64);
    for (int stride = 1; stride < leaves_per_chunk; stride <<= 1) {
      for (int i = tid * 2 * stride; i + stride < nleaf; i += 256 * 2 * stride) s_blk[i] = s_blk[i] + s_blk[i + stride];
      __syncthreads();
    }
    if (tid == 0) {
      double s = 0.0;
      for (int c = 0; c < nleaf; c += leaves_per_chunk) s += s_blk[c];
      s_total = s;
    }
    __syncthreads();
    const double total = s_total;

    // ---- row-major compaction of the psf > 0 cells into products (col*w, row*w) ------------------
    const int nchunks = n / 64;
    for (int ch = wave; ch < nchunks; ch += 4) {
      const double v = raw[ch * 64 + lane];
      const unsigned long long m = __ballot(v > 0.0);
      if (lane == 0) s_cnt[ch] = __popcll(m);
    }
    __syncthreads();
    {
      int base = tid * 4, v[4], s = 0;
      for (int k = 0; k < 4; ++k) { v[k] = (base + k < nchunks) ? s_cnt[base + k] : 0; s += v[k]; }
      int incl = s;
      for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
      __shared__ int s_wtot[4];
      if (lane == 63) s_wtot[wave] = incl;
      __syncthreads();
      int wbase = 0;
      for (int k = 0; k < wave; ++k) wbase += s_wtot[k];
      int excl = wbase + incl - s;
      __syncthreads();
      for (int k = 0; k < 4; ++k) { if (base + k < nchunks) s_cnt[base + k] = excl; excl += v[k]; }
      if (tid == 255) s_cnt[nchunks] = excl;
      __syncthreads();
    }
    const int nnz = s_cnt[nchunks];
    for (int ch = wave; ch < nchunks; ch += 4) {
      const int i = ch * 64 + lane;
      const double v = raw[i];
      const bool nz = v > 0.0;
      const unsigned long long m = __ballot(nz);
      if (nz) {
        const int pos = s_cnt[ch] + __popcll(m & ((1ull << lane) - 1));
        const int r = i / canvas, c = i - r * canvas;
        const double w = v / total;                     // generate_PSF.py:114
        work[pos] = make_double2((double)c * w, (double)r * w);   // :116-117
      }
    }
    __syncthreads();
    __threadfence_block();
    // ---- sequential centroid in list order (one lane), staged through LDS -----------------------
    double ax = 0.0, ay = 0.0;
    for (int base = 0; base < nnz; base += 1024) {
      const int cnt = min(1024, nnz - base);
      for (int i = tid; i < cnt; i += 256) s_stage[i] = work[base + i];
      __syncthreads();
      if (tid == 0) {
        for (int i = 0; i < cnt; ++i) { ax += s_stage[i].x; ay += s_stage[i].y; }
      }
      __syncthreads();
    }
    if (tid == 0) {
      s_off[0] = (int)(ax - (double)canvas / 2);  // offsetX, truncation toward zero (:119)
      s_off[1] = (int)(ay - (double)canvas / 2);  // offsetY (:120)
    }
    __syncthreads();
  } else {
    if (tid == 0) { s_off[0] = 0; s_off[1] = 0; }
    __syncthreads();
  }

  // ---- roll by (-offsetY, -offsetX), crop the centre out_n window, convert -----------------------
  const int ox = s_off[0], oy = s_off[1], c0 = (canvas - out_n) / 2;
  for (int i = tid; i < out_n * out_n; i += 256) {
    const int r = i / out_n, c = i - r * out_n;
    int sr = (r + c0 + oy) % canvas, sc = (c + c0 + ox) % canvas;
    if (sr < 0) sr += canvas;
    if (sc < 0) sc += canvas;
    const double v = raw[sr * canvas + sc];
    if (out64) out64[(size_t)b * out_n * out_n + i] = v;
    if (out16) out16[(size_t)b * out_n * out_n + i] = __float2half_rn(__double2float_rn(v));  // engine.py:84
  }
}

}  // namespace dib

using namespace dib;

// workspace: raw canvases [B][canvas^2] f64 | centroid products [B][canvas^2] double2
extern "C" size_t dib_psf_rasterize_workspace_bytes(int B, int iters, int canvas) {
  (void)iters;
  if (B <= 0 || canvas <= 0) return 0;
  return (size_t)B * canvas * canvas * (sizeof(double) + sizeof(double2));
}

extern "C" int dib_psf_rasterize(const double *traj_dev, int B, int iters, const double *fraction, int canvas,
                                 int center, int out_n, double *psf64_dev, void *psf16_dev, void *workspace_dev,
                                 void *stream) {
  if (B < 0 || (B > 0 && (!traj_dev || !fraction || !workspace_dev))) { set_error("dib_psf_rasterize: null pointer"); return DIB_EINVAL; }
  if (canvas != 64 && canvas != 128 && canvas != 256) { set_error("dib_psf_rasterize: canvas must be 64, 128 or 256, got %d", canvas); return DIB_EINVAL; }
  if (out_n != canvas && !(canvas == 256 && out_n == 128)) { set_error("dib_psf_rasterize: out_n must be canvas, or 128 with canvas 256"); return DIB_EINVAL; }
  if (iters <= 0) { set_error("dib_psf_rasterize: iters must be positive"); return DIB_EINVAL; }
  if (B == 0) return DIB_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)canvas * canvas;
  double *raw = (double *)workspace_dev;
  double2 *work = (double2 *)(raw + (size_t)B * n);
  for (int b0 = 0; b0 < B; b0 += RASTER_CHUNK) {
    FnBatch fnb;
    const int cnt = B - b0 < RASTER_CHUNK ? B - b0 : RASTER_CHUNK;
    // fraction * iters in float64, as `self.fraction[j] * self.iters` (generate_PSF.py:47)
    for (int i = 0; i < cnt; ++i) fnb.fn[i] = fraction[b0 + i] * (double)iters;
    hipLaunchKernelGGL(raster_rows_kernel, dim3(canvas, cnt), dim3(256), 0, s, (const double2 *)traj_dev + (size_t)b0 * iters,
                       iters, fnb, canvas, raw + (size_t)b0 * n);
  }
  DIB_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(raster_finish_kernel, dim3(B), dim3(256), 0, s, raw, canvas, center, out_n, psf64_dev, (__half *)psf16_dev, work);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}
