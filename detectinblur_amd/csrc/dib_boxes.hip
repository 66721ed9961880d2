// Box growth by the PSF extent + clamping, fused into one tiny kernel per image.
// Reference: utils.py:360-392 (`expand_targets`: ~25 launches and several boolean-mask syncs per
// image) and utils.py:395-434 (`fix_bounding_box_squeeze`).  The extents come from the tap table
// header written by dib_psf_compact, so no nonzero()/min()/max() pass runs here.
#include "dib_common.h"

namespace dib {

__device__ inline void clamp_box(float &x1, float &y1, float &x2, float &y2, float wmax, float hmax) {
  // utils.py:398-409 (and again :421-432): upper clamp, then lower clamp, per coordinate
  if (x1 > wmax) x1 = wmax;
  if (y1 > hmax) y1 = hmax;
  if (x2 > wmax) x2 = wmax;
  if (y2 > hmax) y2 = hmax;
  if (x1 < 0.f) x1 = 0.f;
  if (y1 < 0.f) y1 = 0.f;
  if (x2 < 0.f) x2 = 0.f;
  if (y2 < 0.f) y2 = 0.f;
}

__global__ void boxes_kernel(float4 *__restrict__ boxes, int n, const int *__restrict__ tab, int H, int W) {
#pragma clang fp contract(off)
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 b = boxes[i];
  if (tab) {  // utils.py:376-386; 63 = K/2-1 for the only legal K (128)
    b.x += (float)(tab[HDR_CMIN] - 63);
    b.z += (float)(tab[HDR_CMAX] - 63);
    b.y += (float)(tab[HDR_RMIN] - 63);
    b.w += (float)(tab[HDR_RMAX] - 63);
  }
  const float wmax = (float)(W - 1), hmax = (float)(H - 1);
  clamp_box(b.x, b.y, b.z, b.w, wmax, hmax);
  if (b.x >= b.z) { b.z += 1.f; b.x -= 1.f; }  // utils.py:412-414
  if (b.y >= b.w) { b.w += 1.f; b.y -= 1.f; }  // utils.py:416-418
  clamp_box(b.x, b.y, b.z, b.w, wmax, hmax);
  boxes[i] = b;
}

}  // namespace dib

static int launch_boxes(float *boxes_dev, int N, const void *table_dev, int H, int W, void *stream) {
  if (N < 0 || (N > 0 && !boxes_dev)) { dib::set_error("boxes: null pointer or negative count"); return DIB_EINVAL; }
  if (N == 0) return DIB_OK;
  if (((uintptr_t)boxes_dev & 15) != 0) { dib::set_error("boxes: pointer must be 16-byte aligned"); return DIB_EINVAL; }
  hipLaunchKernelGGL(dib::boxes_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, (float4 *)boxes_dev, N,
                     (const int *)table_dev, H, W);
  DIB_HIP_CHECK(hipGetLastError());
  return DIB_OK;
}

extern "C" int dib_expand_boxes(float *boxes_dev, int N, const void *table_dev, int H, int W, void *stream) {
  if (!table_dev) { dib::set_error("dib_expand_boxes: null tap table"); return DIB_EINVAL; }
  return launch_boxes(boxes_dev, N, table_dev, H, W, stream);
}

extern "C" int dib_clamp_boxes(float *boxes_dev, int N, int H, int W, void *stream) {
  return launch_boxes(boxes_dev, N, nullptr, H, W, stream);
}
